# Evidence of the shipped step: bench line, per-kernel stats of the same command (rocprofv3 --kernel-trace --stats),
# two-stream (default) and single-stream.  Everything lands in gpurun_out/$1 (default r03_step).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
O=gpurun_out/${1:-r03_step}; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -3 $O/bench_default.err; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/prof.log 2>&1 || { tail -3 $O/prof.log; exit 1; }
python3 tools/prof_summary.py $O/prof/step_results.db 40 > $O/step_kernel_stats_two_stream.csv && rm -rf $O/prof
export DEVIS_WGRAD_STREAM=0
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/prof1.log 2>&1 || { tail -3 $O/prof1.log; exit 1; }
python3 tools/prof_summary.py $O/prof/step_results.db 40 > $O/step_kernel_stats_single_stream.csv && rm -rf $O/prof
head -c 400 $O/bench_default.json; echo
