"""Launch-by-launch dump of ONE MetaModel outer step from a rocprofv3 --kernel-trace database (the last complete one): everything
between the k_adam before the first k_fd_* / k_meta_* launch of the outer step and its k_meta_sgd.
usage: python tools/kouter.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
sgd = [i for i, r in enumerate(rows) if "k_meta_sgd" in r[0]]
end = sgd[-1]
a = sgd[-2] + 1
# skip the inner steps after the previous outer step: start at the last k_adam before the first k_fd_step_size
first_fd = next(i for i in range(a, end) if "k_fd_" in rows[i][0] or "k_meta_select_bwd" in rows[i][0] or "k_score_dense" in rows[i][0])
start = max([i for i in range(a, first_fd) if rows[i][0].startswith("k_adam")] + [a - 1]) + 1
t0 = rows[start][1]
busy = 0
for i in range(start, end + 1):
    n, s, e = rows[i]
    gap = s - rows[i - 1][2] if i > start else 0
    busy += e - s
    print(f"{i - start:3d} {n[:60]:60s} {(s - t0) / 1e3:9.2f} {(e - s) / 1e3:7.2f} {gap / 1e3:7.2f}")
print(f"# outer step: {end - start + 1} launches, span {(rows[end][2] - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
