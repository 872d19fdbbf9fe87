"""Summarise rocprofv3 --pmc results (rocpd sqlite): per kernel name, mean of each counter per dispatch."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
q = "select k.name, p.counter_name, avg(p.counter_value), count(*) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name"
try:
    rows = c.execute(q).fetchall()
except Exception as e:
    print(cols); raise
d = collections.defaultdict(dict)
for name, cn, v, n in rows:
    d[name][cn] = v
flt = sys.argv[2:] 
for name, cs in d.items():
    if flt and not any(f in name for f in flt): continue
    print(name[:60])
    print("   " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(cs.items())))
