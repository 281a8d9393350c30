run() { echo "== $*"; env "$@" timeout -k 10 120 python3 tools/bench_post.py 2>&1 | grep launch | tail -1 || exit 1; }
L=isaaclab_amd/_lib
run MDP_HIP_LIB=$L/libmdp_hip.so
run MDP_HIP_LIB=$L/libmdp_hip.so MDP_POST_THREADS=256 MDP_POST_CAST_SLOT_W=64
run MDP_HIP_LIB=$L/libmdp_hip_w5.so MDP_POST_THREADS=256 MDP_POST_CAST_SLOT_W=64
run MDP_HIP_LIB=$L/libmdp_hip_w4.so MDP_POST_THREADS=256 MDP_POST_CAST_SLOT_W=64
run MDP_HIP_LIB=$L/libmdp_hip_w5.so MDP_POST_THREADS=256 MDP_POST_CAST_SLOT_W=128
run MDP_HIP_LIB=$L/libmdp_hip.so MDP_POST_THREADS=192 MDP_POST_CAST_SLOT_W=64
run MDP_HIP_LIB=$L/libmdp_hip.so MDP_POST_THREADS=192 MDP_POST_CAST_SLOT_W=96
run MDP_HIP_LIB=$L/libmdp_hip_w5.so MDP_POST_THREADS=192 MDP_POST_CAST_SLOT_W=96
run MDP_HIP_LIB=$L/libmdp_hip_w5.so MDP_POST_THREADS=512 MDP_POST_CAST_SLOT_W=64
run MDP_HIP_LIB=$L/libmdp_hip_w5.so
