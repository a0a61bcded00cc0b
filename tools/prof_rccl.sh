# kernel trace of the one-rank RCCL rehearsal, then what runs around each collective (tools/rccl_windows.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 DEVIS_BENCH_FORCE_DIST=1
O=gpurun_out/rccl_trace; rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace -d $O/prof -o step -- python3 bench.py --steps 12 --warmup 6 --no-cpu-baseline > $O/prof.log 2>&1 || { tail -3 $O/prof.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*' $O/prof.log
python3 tools/rccl_windows.py $O/prof/step_results.db 120 > $O/windows.txt && rm -rf $O/prof
