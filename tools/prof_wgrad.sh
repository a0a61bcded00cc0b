# kernel-trace durations of tools/time_wgrad_c16.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/prof_wg && timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_wg -o c -- python3 tools/time_wgrad_c16.py > gpurun_out/prof_wg.log 2>&1 || { tail -3 gpurun_out/prof_wg.log; exit 1; }
tail -2 gpurun_out/prof_wg.log | cut -c1-120
python3 tools/prof_summary.py gpurun_out/prof_wg/c_results.db 1 | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if 'wgrad3' in r[0] or 'reduce' in r[0]: print('   %-60s n %4s %7.1f us' % (r[0][12:72], r[1], float(r[3])/1e3))
"
