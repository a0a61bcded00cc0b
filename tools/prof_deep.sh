#!/bin/bash
# kernel-trace durations of the 64- / 128-channel 3x3x3 layers (tools/time_deep.py calls them as the network does)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/prof_deep && timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_deep -o c -- python3 tools/time_deep.py > gpurun_out/prof_deep.log 2>&1 || { tail -3 gpurun_out/prof_deep.log; exit 1; }
python3 tools/prof_summary.py gpurun_out/prof_deep/c_results.db 1 | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if 'k_conv3_deep' in r[0]: print('   %-70s n %4s %7.1f us' % (r[0][12:82], r[1], float(r[3])/1e3))
"
rm -rf gpurun_out/prof_deep
