# the same seeded run twice per setting: final losses must agree to the last digit
export TMPDIR=/tmp
l3() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('3-D final_loss %.9f' % json.loads(sys.stdin.read())['final_loss'])"; }
echo "two streams:"; l3 && l3 && echo "one stream:" && DEVIS_WGRAD_STREAM=0 l3 && DEVIS_WGRAD_STREAM=0 l3
