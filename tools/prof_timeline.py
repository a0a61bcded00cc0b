#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace .db: per steady-state window, wall span, union of kernel intervals (GPU busy),
sum of kernel durations, and the largest idle gaps with the kernels around them."""
import sqlite3
import sys

db = sys.argv[1]
skip_frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4     # ignore the warm-up part of the run
c = sqlite3.connect(db)
rows = c.execute("select start, end, name, queue_id from kernels order by start").fetchall() if \
    "queue_id" in [r[1] for r in c.execute("pragma table_info(kernels)")] else \
    [(r[0], r[1], r[2], 0) for r in c.execute("select start, end, name from kernels order by start")]
# steady-state window: from the end of the 8th optimizer launch to the end of the last one
adam = [i for i, r in enumerate(rows) if "k_adam" in r[2]]
if len(adam) >= 12:
    nsteps = len(adam) - 8
    rows = rows[adam[7] + 1: adam[-1] + 1]
else:
    nsteps = 1
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + (t1 - t0) * skip_frac
    rows = [r for r in rows if r[0] >= cut]
print("steps in window:", nsteps)
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, name, q in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in rows)
print("per step: window %.3f ms: GPU busy %.3f ms (%.1f %%), sum of kernel durations %.3f ms, %.0f launches, %.0f idle gaps totalling %.3f ms"
      % (span / 1e6 / nsteps, busy / 1e6 / nsteps, 100.0 * busy / span, tot / 1e6 / nsteps, len(rows) / nsteps,
         len(gaps) / nsteps, (span - busy) / 1e6 / nsteps))
queues = {}
for s, e, name, q in rows:
    queues[q] = queues.get(q, 0) + (e - s)
print("per queue (ms):", {k: round(v / 1e6, 2) for k, v in queues.items()})
gaps.sort(reverse=True)
for g, name in gaps[:12]:
    print("  gap %.1f us before %s" % (g / 1e3, name[:90]))
import collections
hist = collections.Counter(min(int(g / 1000), 20) for g, _ in gaps)
print("gap histogram (us: count):", dict(sorted(hist.items())))
