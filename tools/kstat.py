"""Per-kernel summary (count / avg / min / max / share) of a rocprofv3 --kernel-trace rocpd sqlite database.
usage: python tools/kstat.py <results.db> [top_n]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                 "from kernels group by name order by 6 desc").fetchall()
tot = sum(r[5] for r in rows)
print(f"# rocprofv3 --kernel-trace summary of {sys.argv[1]}; total kernel time {tot / 1e6:.3f} ms")
print(f"{'kernel':58s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share%':>7s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{r[0][:58]:58s} {r[1]:6d} {r[2] / 1e3:9.2f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} {100 * r[5] / tot:7.2f}")
