"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (rocpd sqlite).

Counters expected in the pass: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES.
Per dispatch the per-instance samples (one per SE/XCC dimension) are summed (max for GRBM_GUI_ACTIVE), then averaged over the
dispatches of a kernel.  MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * 1024 SIMDs)  — rocprofv3's own formula
(`rocprofv3 -L`: MfmaUtil);  TF/s(F32) = MOPS_F32 * 512 / (GRBM_GUI_ACTIVE / clock)."""
import collections
import json
import sqlite3
import sys

SIMDS = 256 * 4
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2:]
rows = db.execute("select k.name, p.dispatch_id, p.counter_name, sum(p.counter_value), max(p.counter_value), count(*) "
                  "from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id group by p.dispatch_id, p.counter_name").fetchall()
per = collections.defaultdict(lambda: collections.defaultdict(list))
for name, did, cn, s, mx, n in rows:
    per[name][cn].append(mx if cn == "GRBM_GUI_ACTIVE" else s)
out = {}
for name, cs in per.items():
    if flt and not any(f in name for f in flt):
        continue
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    nd = len(next(iter(cs.values())))
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    rec = {"dispatches": nd, **{k: round(v, 1) for k, v in m.items()}}
    if gui > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        rec["MfmaUtil_pct"] = round(100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS), 2)
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
        rec["mfma_f32_flops_per_dispatch"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
    out[name[:70]] = rec
print(json.dumps(out, indent=1))
