#!/usr/bin/env python3
"""Average the counters of a rocprofv3 --pmc run per kernel: tools/pmc_summary.py <output dir> <kernel-name substring>...
-> CSV lines kernel, counter, dispatches, mean value, mean duration (us)."""
import csv
import glob
import os
import sys
from collections import defaultdict

d, subs = sys.argv[1], sys.argv[2:]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if subs and not any(s in name for s in subs):
            continue
        dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3 if r.get("End_Timestamp") else 0.0
        acc[(name[:100], r["Counter_Name"])].append((float(r["Counter_Value"]), dur))
print("kernel,counter,dispatches,mean_value,mean_us")
for (k, c), v in sorted(acc.items()):
    print('"%s",%s,%d,%.1f,%.2f' % (k, c, len(v), sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v)))
