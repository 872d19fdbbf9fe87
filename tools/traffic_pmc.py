"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes).
usage: python tools/traffic_pmc.py <fetch.db> <write.db> > out.json
Per dispatch the per-instance samples are summed, then averaged over a kernel's dispatches.  Units: rocprofv3 reports KB.
gfx950 correction (guide, HBM section): FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads -> x2."""
import collections
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select k.name, p.dispatch_id, sum(p.counter_value) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id "
                     "where p.counter_name = ? group by p.dispatch_id", (counter,)).fetchall()
    acc = collections.defaultdict(list)
    for name, _, v in rows:
        acc[name].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for name in sorted(set(f) | set(w)):
    if "k_" not in name or "at::native" in name:
        continue
    fk, n = f.get(name, (0.0, 0))
    wk, _ = w.get(name, (0.0, 0))
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    out[short] = {"dispatches": n, "FETCH_SIZE_KB_raw": round(fk, 2), "WRITE_SIZE_KB": round(wk, 2),
                  "hbm_bytes_per_launch": round((2.0 * fk + wk) * 1024.0)}
print(json.dumps(out, indent=1))
