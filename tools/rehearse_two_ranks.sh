# Two-rank rehearsal of bench.py's N > 1 control flow on the one-GPU box (gloo transport, both ranks on cuda:0): the
# shipped host-staged statistics all-reduce, then (diagnosis) gloo's device-tensor path under a 60 s watchdog that dumps
# where the ranks stall.  bash tools/rehearse_two_ranks.sh   ->   gpurun_out/r2b_rehearse_{host,dev}.log
cd $GRAFT_REPO_ROOT
export MDP_BENCH_REHEARSAL=1 MDP_BENCH_TRACE=1
echo "== host-staged gloo (shipped path)"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --regions 5 --no-cpu-baseline > gpurun_out/r2b_rehearse_host.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r2b_rehearse_host.log | cut -c1-600
echo "== gloo device-tensor path (diagnosis, watchdog 60 s)"
MDP_GLOO_DEVICE_PATH=1 MDP_BENCH_WATCHDOG=60 timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 20 --warmup 5 --regions 5 --no-cpu-baseline > gpurun_out/r2b_rehearse_dev.log 2>&1; echo "rc=$?"; tail -60 gpurun_out/r2b_rehearse_dev.log | cut -c1-300
