"""durations (us) of every launch between two consecutive k_adam launches in the middle of a rocprofv3 kernel-trace db"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if r[0].startswith("k_adam")]
a, b = ends[len(ends) // 2], ends[len(ends) // 2 + 1]
prev = rows[a][2]
for name, st, en in rows[a + 1:b + 1]:
    print("%-60s gap %6.2f dur %7.2f" % (name[:60], (st - prev) / 1e3, (en - st) / 1e3))
    prev = en
print("step period %.1f us" % ((rows[b][2] - rows[a][2]) / 1e3))
