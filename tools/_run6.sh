cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r2d_t1.log 2>&1 || { tail -40 gpurun_out/r2d_t1.log; exit 1; }
tail -3 gpurun_out/r2d_t1.log
bash tools/profile_round.sh r02d > gpurun_out/r02d_profile.log 2>&1; tail -8 gpurun_out/r02d_profile.log | cut -c1-300
bash tools/measure_configs.sh r02d > gpurun_out/r02d_measure.log 2>&1; tail -14 gpurun_out/r02d_measure.log | cut -c1-200
MDP_HIP_LIB=isaaclab_amd/_lib/libmdp_hip_stamps.so timeout -k 10 200 python -u tools/kernel_stamps.py > gpurun_out/r02d_stamps.txt 2>&1; head -12 gpurun_out/r02d_stamps.txt
