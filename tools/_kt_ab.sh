O=gpurun_out
for v in old main; do
  lib=isaaclab_amd/_lib/libmdp_hip_$v.so; [ "$v" = main ] && lib=isaaclab_amd/_lib/libmdp_hip.so
  export MDP_HIP_LIB=$lib
  rm -rf $O/kt_$v
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt_$v -o kt -- python3 bench.py --no-cpu-baseline --regions 20 > $O/kt_$v.log 2>&1
  echo "== $v"; python3 tools/rocprof_summary.py $(find $O/kt_$v -name "*_results.db" | head -1) 2>&1 | head -4
  find $O/kt_$v -name "*.db" -delete
done
