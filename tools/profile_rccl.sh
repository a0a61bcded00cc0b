# kernel trace of the one-rank RCCL rehearsal (env:// rendezvous, no launcher between rocprofv3 and python)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/prof_rccl
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 DEVIS_BENCH_FORCE_DIST=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rccl -o rc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/rc.log 2>&1 && python tools/prof_summary.py gpurun_out/prof_rccl/rc_results.db 25 > gpurun_out/rc_stats.csv && python tools/prof_timeline.py gpurun_out/prof_rccl/rc_results.db > gpurun_out/rc_timeline.txt 2>&1
