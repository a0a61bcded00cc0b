# one-rank rehearsal of the RCCL gradient all-reduce path (the real N>1 runs are the driver's), then the same under
# torch.distributed.run without collectives, then plain: three lines to compare on ONE box
export TMPDIR=/tmp
DEVIS_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | cut -c1-260 &&
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep metric | cut -c1-260 &&
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | cut -c1-260
