#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace .db (tools/prof_timeline.sh): the kernels of the last 0.6 ms before an optimizer launch
of a steady-state step - start / end relative to the k_adam start, queue, name - i.e. what the step's tail consists of."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else "0"
rows = c.execute("select start, end, name, %s from kernels order by start" % q).fetchall()
adam = [i for i, r in enumerate(rows) if "k_adam" in r[2]]
i = adam[-2]
t0 = rows[i][0]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
for s, e, name, qq in rows[:i + 1]:
    if e >= t0 - win * 1e3 and s <= t0:
        print("%8.1f %8.1f  q%-3s %6.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, qq, (e - s) / 1e3, name[:90]))
