# kernel-trace durations of one bench step's DeUp_Cat kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/prof_up && timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_up -o c -- python3 bench.py --steps 5 --warmup 3 > gpurun_out/prof_up.log 2>&1 || { tail -3 gpurun_out/prof_up.log; exit 1; }
python3 tools/prof_summary.py gpurun_out/prof_up/c_results.db 1 | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if 'upcat' in r[0] or 'k_pw_' in r[0] or 'TOTAL' in r[0]: print('   %-70s n %4s %7.1f us' % (r[0][12:82], r[1], float(r[3] or 0)/1e3))
"
