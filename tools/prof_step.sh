# r02 evidence of the shipped step: bench line, per-kernel stats of the same command (rocprofv3 --kernel-trace --stats),
# per-step series and phase times.  Everything lands in gpurun_out/r02_final/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
O=gpurun_out/r02_final; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -3 $O/bench_default.err; exit 1; }
timeout -k 10 100 python3 tools/step_series.py 100 > $O/step_series.txt 2>/dev/null || exit 1
DEVIS_LAUNCH_PLAN=0 timeout -k 10 100 python3 tools/step_series.py 100 > $O/step_series_eager.txt 2>/dev/null || exit 1
timeout -k 10 100 python3 tools/phase_times.py > $O/phase_times.txt 2>/dev/null || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/prof.log 2>&1 || { tail -3 $O/prof.log; exit 1; }
python3 tools/prof_summary.py $O/prof/step_results.db 40 > $O/step_kernel_stats.csv && rm -rf $O/prof
head -c 600 $O/bench_default.json; echo; cat $O/step_series.txt | head -2; cat $O/step_series_eager.txt | head -2; cat $O/phase_times.txt
