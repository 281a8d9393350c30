# Same-box A/B of library variants x manager-block sizes on the two big launches (GPU box):
#   bash tools/ab_post.sh "old:24 main:24 w6:12 ..."     (variant "main" = the shipped library)
O=gpurun_out
mkdir -p $O
for spec in $1; do
  v=${spec%%:*}; e=${spec##*:}
  lib=isaaclab_amd/_lib/libmdp_hip_$v.so
  [ "$v" = main ] && lib=isaaclab_amd/_lib/libmdp_hip.so
  echo "== $v E=$e $2"
  env MDP_HIP_LIB=$lib MDP_POST_ENVS=$e $2 timeout -k 10 120 python3 tools/bench_post.py 2>&1 | grep launch | tail -2 || exit 1
done
