set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r2b_t2.log 2>&1 || { tail -40 gpurun_out/r2b_t2.log; exit 1; }
tail -3 gpurun_out/r2b_t2.log
bash tools/ab_post.sh "old:24 main:24 w6:24 w6:16 w8:12" 2>&1 | tee gpurun_out/r2b_ab2.log
bash tools/ab_post.sh "old:24 w6:24" "MDP_TERRAIN=synthetic" 2>&1 | tee -a gpurun_out/r2b_ab2.log
MDP_HIP_LIB=isaaclab_amd/_lib/libmdp_hip_stamps.so timeout -k 10 120 python3 tools/kernel_stamps.py > gpurun_out/r2b_stamps2.log 2>&1; tail -16 gpurun_out/r2b_stamps2.log
