# SQ counters of the cast / post-physics launches for library variants: bash tools/pmc_variants.sh "main aw1 aw2"
O=gpurun_out/pmcv; rm -rf $O; mkdir -p $O
for v in $1; do
  lib=isaaclab_amd/_lib/libmdp_hip_$v.so; [ "$v" = main ] && lib=isaaclab_amd/_lib/libmdp_hip.so
  export MDP_HIP_LIB=$lib
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES -d $O/$v -o p -- python3 tools/prof_cast.py > $O/$v.log 2>&1
  echo "== $v"; python3 tools/rocprof_pmc.py $(find $O/$v -name "*_results.db" | head -1) 2>&1 | grep -E "k_raycaster_update_down|k_post_physics" | cut -c1-60,80-400
  timeout -k 10 120 python3 tools/bench_post.py 2>&1 | grep launch | tail -1
done
find $O -name "*.db" -delete
