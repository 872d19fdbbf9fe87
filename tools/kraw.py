"""raw rows of one steady-state step from a rocprofv3 kernel-trace db (all columns that identify queue / stream / dispatch)"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
print(cols)
rows = c.execute("select * from kernels order by start").fetchall()
ix = {n: i for i, n in enumerate(cols)}
starts = [i for i, r in enumerate(rows) if r[ix["name"]].startswith("k_prep")]
a = starts[len(starts) // 2]
prev_end = rows[a - 1][ix["end"]]
keep = [n for n in cols if n not in ("name", "start", "end")]
for r in rows[a - 2:a + 20]:
    print(r[ix["name"]][:28].ljust(28), "gap %7.2f dur %8.2f" % ((r[ix["start"]] - prev_end) / 1e3, (r[ix["end"]] - r[ix["start"]]) / 1e3),
          {n: r[ix[n]] for n in keep if n in ("queue_id", "stream_id", "dispatch_id", "tid", "agent_abs_index", "lds_size", "scratch_size", "grid_x", "workgroup_x", "grid_size_x", "workgroup_size_x")})
    prev_end = r[ix["end"]]
print("--- per-step gaps (us) after k_prep and before the 3rd launch, over 15 consecutive steps")
for a in starts[len(starts) // 2: len(starts) // 2 + 15]:
    g0 = (rows[a][ix["start"]] - rows[a - 1][ix["end"]]) / 1e3
    g1 = (rows[a + 1][ix["start"]] - rows[a][ix["end"]]) / 1e3
    g2 = (rows[a + 2][ix["start"]] - rows[a + 1][ix["end"]]) / 1e3
    print("before prep %8.2f  after prep %8.2f  before 3rd %8.2f   dispatch %d" % (g0, g1, g2, rows[a][ix["dispatch_id"]]))
