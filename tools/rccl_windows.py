#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace .db of the data-parallel bench: what runs around every RCCL kernel of one steady-state
step (start / end relative to the collective's start, queue, name), and the time no kernel at all was running in the
window.  python tools/rccl_windows.py trace.db [window_us]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else "0"
rows = c.execute("select start, end, name, %s from kernels order by start" % q).fetchall()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
adam = [i for i, r in enumerate(rows) if "k_adam" in r[2]]
lo, hi = adam[-3], adam[-2]
for i in range(lo, hi + 1):
    s0, e0, name, _ = rows[i]
    if "Reduce" not in name and "nccl" not in name.lower() and "rccl" not in name.lower():
        continue
    print("---- %s  (%.1f us, %.0f us before k_adam)" % (name[:60], (e0 - s0) / 1e3, (rows[hi][0] - s0) / 1e3))
    w = [(s, e, n, qq) for s, e, n, qq in rows[lo:hi + 1] if e >= s0 - win * 1e3 and s <= e0 + win * 1e3]
    for s, e, n, qq in w:
        print("%8.1f %8.1f  q%-3s %6.1f us  %s" % ((s - s0) / 1e3, (e - s0) / 1e3, qq, (e - s) / 1e3, n[:80]))
    # idle time inside the window
    ev = sorted((max(s, s0 - win * 1e3), min(e, e0 + win * 1e3)) for s, e, _, _ in w)
    cur, idle = s0 - win * 1e3, 0.0
    for s, e in ev:
        if s > cur:
            idle += s - cur
        cur = max(cur, e)
    idle += max(0.0, e0 + win * 1e3 - cur)
    print("     no kernel running for %.1f us of the %.0f us window" % (idle / 1e3, (e0 - s0) / 1e3 + 2 * win))
