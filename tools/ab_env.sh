# A/B of one environment switch on one box: tools/ab_env.sh VAR [rounds]; bench.py --steps 200, alternating
cd $GRAFT_REPO_ROOT || exit 1
VAR=$1; R=${2:-3}
for i in $(seq 1 $R); do
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; print('default ', json.loads(sys.stdin.read())['ms_per_step'])" || exit 1
  env $VAR=1 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; print('$VAR=1', json.loads(sys.stdin.read())['ms_per_step'])" || exit 1
done
