#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (kernel-trace) as a kernel_stats-style CSV: name, calls, total ns, avg ns, %."""
import sqlite3
import sys

db = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else cols[0]
rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name, name)).fetchall()
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","TotalPerStepUs"')
for r in rows:
    print('"%s",%d,%d,%.1f,%.2f,%d,%d,%.1f' % (r[0][:150], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5], r[2] / steps / 1e3))
print('"TOTAL",%d,%d,,,,,%.1f' % (sum(r[1] for r in rows), tot, tot / steps / 1e3))
