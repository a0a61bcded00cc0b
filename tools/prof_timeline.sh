# kernel trace of the bench step, then tools/timeline.py: which kernels of the two streams overlap, per phase
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/prof_tl && timeout -s KILL 200 rocprofv3 --kernel-trace -d gpurun_out/prof_tl -o c -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/prof_tl.log 2>&1 || { tail -3 gpurun_out/prof_tl.log; exit 1; }
python3 tools/timeline.py gpurun_out/prof_tl/c_results.db
