# rocprofv3 evidence for the fused evidential head at 2 x 128^3, K = 4 (tools/bench_head.py): kernel-trace durations,
# then one counter pass per run (SQ instruction / cycle counters; FETCH_SIZE; WRITE_SIZE; GRBM clock).
# Output: gpurun_out/head_*.csv.  Every pass has its own kill timer; a failed pass ends the script.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
O=gpurun_out/prof_head
rm -rf $O && mkdir -p $O
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/trace -o head -- python3 tools/bench_head.py --reps 10 > $O/trace.log 2>&1 || { echo trace failed; tail -5 $O/trace.log; exit 1; }
python3 tools/prof_summary.py $O/trace/head_results.db 1 > gpurun_out/head_kernel_stats.csv || exit 1
grep -E "dce4|crit|finalize" gpurun_out/head_kernel_stats.csv | cut -c1-200
pass() {  # name, counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 tools/bench_head.py --reps 2 > $O/$n.log 2>&1 || { echo pass $n failed; tail -5 $O/$n.log; exit 1; }
  python3 tools/pmc_summary.py $O/$n dce4 finalize > gpurun_out/head_pmc_$n.csv || exit 1
  cat gpurun_out/head_pmc_$n.csv
}
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass clk GRBM_GUI_ACTIVE
