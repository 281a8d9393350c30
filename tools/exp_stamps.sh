for w in 0 1; do
  echo "### MDP_EXP_DROP_WALLS=$w"
  if [ $w = 1 ]; then export MDP_EXP_DROP_WALLS=1; else unset MDP_EXP_DROP_WALLS; fi
  MDP_HIP_LIB=isaaclab_amd/_lib/libmdp_hip_stamps.so timeout -k 10 200 python3 tools/kernel_stamps.py 2>&1 | grep -v amdgpu.ids | head -22 || exit 1
done
