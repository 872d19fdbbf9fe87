"""Timeline of ONE steady-state training step from a rocprofv3 --kernel-trace database: per kernel start offset, duration
and the idle gap before it (MEDIAN over the last `n` steps: eager warm-up steps and graph-launch boundaries do not smear in).  A step = the launches after one k_adam up to the next k_adam; the most frequent launch count is shown (steps that carry
their own k_prep — the first of each graph — have one more).
usage: python tools/ktimeline.py <results.db> [n_steps]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# a step ends with its optimizer launch (k_adam); since dr4sr_sasrec_train_steps only the first step of a graph has its own k_prep
ends = [i for i, r in enumerate(rows) if r[0].startswith("k_adam") or r[0].startswith("void k_adam")]
steps = [(ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]
lens = {}
for a, b in steps:
    if b - a > 1:
        lens[b - a] = lens.get(b - a, 0) + 1
L = max(lens, key=lens.get)                      # the graph-replayed training step (most frequent multi-launch period)
steps = [(a, b) for a, b in steps if b - a == L][-n_steps:]
import statistics
cols = [[[], [], []] for _ in range(L)]
periods = []
for a, b in steps:
    t0 = rows[a][1]
    for k in range(L):
        name, st, en = rows[a + k]
        prev_end = rows[a + k - 1][2] if k else st
        cols[k][0].append(st - t0)
        cols[k][1].append(en - st)
        cols[k][2].append(st - prev_end)
    periods.append(rows[min(b, len(rows) - 1)][1] - t0)
n = 1
acc = [[statistics.median(c[0]), statistics.median(c[1]), statistics.median(c[2])] for c in cols]
total = statistics.mean(periods)
print(f"# {len(steps)} steps of {L} launches; mean step period {total / 1e3:.2f} us (includes graph-launch boundaries), median {statistics.median(periods) / 1e3:.2f} us")
print(f"{'#':>2s} {'kernel':52s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s}")
busy = 0.0
for k in range(L):
    name = rows[steps[0][0] + k][0]
    print(f"{k:2d} {name[:52]:52s} {acc[k][0] / n / 1e3:9.2f} {acc[k][1] / n / 1e3:8.2f} {acc[k][2] / n / 1e3:7.2f}")
    busy += acc[k][1] / n
print(f"# sum of kernel durations {busy / 1e3:.2f} us, idle inside the step {(total / n - busy) / 1e3:.2f} us")
