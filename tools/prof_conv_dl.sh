# kernel-trace durations of tools/time_conv_c16.py for several D-chunk lengths of the marching kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
for dl in 0 22 16 26; do
  export DEVIS_MARCH_DL=$dl
  rm -rf gpurun_out/prof_dl && timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dl -o c -- python3 tools/time_conv_c16.py > gpurun_out/prof_dl.log 2>&1 || { tail -3 gpurun_out/prof_dl.log; exit 1; }
  echo "DL=$dl"; python3 tools/prof_summary.py gpurun_out/prof_dl/c_results.db 1 | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin):
    if 'k_conv3_c16' in r[0]: print('   %-50s %7.1f us' % (r[0][30:80], float(r[3])/1e3))
"
done
