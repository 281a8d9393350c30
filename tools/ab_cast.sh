# same-box A/B of library variants on the stand-alone cast and the post-physics launch: tools/ab_cast.sh "" _X _Y
for v in "$@"; do
echo "lib=libmdp_hip$v.so"; MDP_RC_BATCH=${MDP_RC_BATCH:-1} MDP_HIP_LIB=isaaclab_amd/_lib/libmdp_hip$v.so timeout -k 10 120 python tools/bench_post.py 2>/dev/null | tail -1 || exit 1
done
