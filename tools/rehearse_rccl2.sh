export TMPDIR=/tmp RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 DEVIS_BENCH_FORCE_DIST=1
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
echo "as is:"; MASTER_PORT=29521 run &&
echo "side stream created before the process group:"; MASTER_PORT=29522 DEVIS_BENCH_PRESTREAM=1 run &&
echo "side stream priority -1:"; MASTER_PORT=29523 DEVIS_WGRAD_STREAM_PRIORITY=-1 run &&
echo "no side stream:"; MASTER_PORT=29524 DEVIS_WGRAD_STREAM=0 run &&
echo "no dist:"; DEVIS_BENCH_FORCE_DIST=0 run
