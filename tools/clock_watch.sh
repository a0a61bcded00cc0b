# log GPU clocks / power while bench.py runs a few times (is the run-to-run spread a clock state?)
cd $GRAFT_REPO_ROOT || exit 1
( for i in $(seq 1 200); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" | tr -s ' ' | tr '\n' '|' ; echo " t=$(date +%s.%N | cut -c1-14)"; sleep 0.15; done ) > gpurun_out/clock_watch.log 2>&1 &
WATCH=$!
for i in 1 2 3 4; do
  echo "run $i start $(date +%s.%N | cut -c1-14)"
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; print('ms', json.loads(sys.stdin.read())['ms_per_step'])" || break
  echo "run $i end $(date +%s.%N | cut -c1-14)"
done
kill $WATCH
wait $WATCH 2>/dev/null
wc -l gpurun_out/clock_watch.log
