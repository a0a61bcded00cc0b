# repeated runs of bench.py --steps 200 under one environment setting: tools/ab_env2.sh "VAR=VAL ..." [runs]
cd $GRAFT_REPO_ROOT || exit 1
R=${2:-4}
for i in $(seq 1 $R); do
  env $1 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; print('$1', json.loads(sys.stdin.read())['ms_per_step'])" || exit 1
done
