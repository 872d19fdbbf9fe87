"""Per launch slot of the modal training step: median / mean / max duration over the steps of a rocprofv3 --kernel-trace database —
shows bimodal launches that a median timeline hides.   usage: python tools/kslots.py <results.db> [name filter]"""
import sqlite3
import statistics
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("select name, start, end from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if r[0].startswith("k_adam")]
steps = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]
lens = {}
for a, b in steps:
    lens[b - a] = lens.get(b - a, 0) + 1
L = max(lens, key=lens.get)
steps = [(a, b) for a, b in steps if b - a == L]
for k in range(L):
    name = rows[steps[0][0] + k][0]
    if flt not in name:
        continue
    d = [(rows[a + k][2] - rows[a + k][1]) / 1e3 for a, b in steps]
    print(f"{k:3d} {name[:48]:48s} median {statistics.median(d):7.2f} mean {statistics.mean(d):7.2f} max {max(d):7.2f}  >2x median: {sum(x > 2 * statistics.median(d) for x in d)}/{len(d)}")
