# one-rank rehearsal of the RCCL gradient all-reduce path (the real N>1 runs are the driver's) on ONE box:
# with the process group and the bucketed all-reduce, the same without the side stream, and the plain step
export TMPDIR=/tmp RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 DEVIS_BENCH_FORCE_DIST=1
run() { timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
echo "one-rank process group, bucketed all-reduce:"; MASTER_PORT=29521 run &&
echo "side stream created before the process group:"; MASTER_PORT=29522 DEVIS_BENCH_PRESTREAM=1 run &&
echo "no side stream:"; MASTER_PORT=29524 DEVIS_WGRAD_STREAM=0 run &&
echo "no process group:"; DEVIS_BENCH_FORCE_DIST=0 run
