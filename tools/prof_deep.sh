# kernel-trace durations of tools/time_deep.py: (2,2,16) tiles vs (2,4,16) tiles of the 128-channel kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
for big in 0 1; do
  if [ $big = 1 ]; then export DEVIS_DEEP128_SMALL_TILE=1; fi
  rm -rf gpurun_out/prof_deep && timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_deep -o c -- python3 tools/time_deep.py > gpurun_out/prof_deep.log 2>&1 || { tail -3 gpurun_out/prof_deep.log; exit 1; }
  echo "small_tile=$big"; python3 tools/prof_summary.py gpurun_out/prof_deep/c_results.db 1 | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if 'k_conv3_deep' in r[0]: print('   %-60s n %4s %7.1f us' % (r[0][12:72], r[1], float(r[3])/1e3))
"
done
