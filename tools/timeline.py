#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 kernel-trace .db: per stream busy time, gaps, the tail the side
stream runs alone, and the main stream's kernels with their stretch against the single-stream durations.
usage: timeline.py results.db [step_index]"""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select start, end, stream_id, name, grid_x / workgroup_x from kernels order by start").fetchall()
heads = [i for i, r in enumerate(rows) if "ncx_to_nxc" in r[3]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(heads) - 3
a, b = heads[k], heads[k + 1]
ev = rows[a:b]
t0, t1 = ev[0][0], rows[b][0]
print("step %d: %.1f us from first kernel to the next step's first kernel, %d kernels" % (k, (t1 - t0) / 1e3, len(ev)))
streams = sorted({r[2] for r in ev})
main = max(streams, key=lambda s: sum(1 for r in ev if r[2] == s))


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


for s in streams:
    iv = [(r[0], r[1]) for r in ev if r[2] == s]
    print("  stream %d%s: %d kernels, sum %.1f us, busy (union) %.1f us, first %.1f last end %.1f" %
          (s, " (main)" if s == main else "", len(iv), sum(e - st for st, e in iv) / 1e3, union(iv) / 1e3,
           (min(st for st, _ in iv) - t0) / 1e3, (max(e for _, e in iv) - t0) / 1e3))
print("  all streams busy (union): %.1f us" % (union([(r[0], r[1]) for r in ev]) / 1e3))
m = [r for r in ev if r[2] == main]
gaps = [(m[i + 1][0] - m[i][1]) / 1e3 for i in range(len(m) - 1)]
print("  main stream gaps: sum %.1f us, > 5 us: %d (sum %.1f)" % (sum(g for g in gaps if g > 0), sum(1 for g in gaps if g > 5),
                                                                   sum(g for g in gaps if g > 5)))
big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:12]
for g, i in big:
    print("     gap %.1f us after %s (at %.0f us)" % (g, m[i][3][7:60], (m[i][1] - t0) / 1e3))
side = [r for r in ev if r[2] != main]
if side:
    main_end = max(r[1] for r in m)
    print("  side stream work after the main stream's last kernel ended: %.1f us" %
          (sum(min(r[1], t1) - max(r[0], main_end) for r in side if r[1] > main_end) / 1e3))
