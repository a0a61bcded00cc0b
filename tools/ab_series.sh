# step_series.py (100 steps, per-step GPU times) under a list of environment settings, one line each:
#   tools/ab_series.sh "A=1" "B=2 C=3" ...     ("-" = no setting)
cd $GRAFT_REPO_ROOT || exit 1
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  env $e timeout -k 10 100 python tools/step_series.py 100 2>/dev/null > gpurun_out/ab_series_tmp.txt || exit 1
  echo "[$cfg] $(sed -n 2p gpurun_out/ab_series_tmp.txt | cut -c1-110) | $(sed -n 1p gpurun_out/ab_series_tmp.txt | cut -c24-60)"
done
