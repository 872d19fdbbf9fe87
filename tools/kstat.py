import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
rows=c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc").fetchall()
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 12]:
    print(f"{r[0][:48]:48s} n={r[1]:5d} avg={r[2]/1e3:8.2f}us")
