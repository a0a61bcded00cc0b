# rocprofv3 counter passes over the bench step (single stream: one kernel at a time, clean per-kernel counters).
# One counter group per run (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Output:
# gpurun_out/pmc_step/<pass>/..., merged per kernel into gpurun_out/pmc_step.json by tools/pmc_step_summary.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
export DEVIS_WGRAD_STREAM=0
O=gpurun_out/pmc_step
rm -rf $O && mkdir -p $O
pass() {
  n=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/$n.log 2>&1 || { echo pass $n failed; tail -5 $O/$n.log; exit 1; }
  echo "pass $n done"
}
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE &&
python3 tools/pmc_step_summary.py $O > gpurun_out/pmc_step.json && python3 -c "
import json
d = json.load(open('gpurun_out/pmc_step.json'))
for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['us_per_step'])[:30]:
    print('%-70s n/step %5.1f us %7.1f  HBM GB/s %7.0f  mfma_util %5.3f' % (k[:70], v['launches_per_step'], v['mean_us'], v['hbm_GBps'], v['mfma_util']))
"
