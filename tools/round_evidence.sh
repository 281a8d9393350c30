# Evidence run of a round (GPU box): bench line + kernel trace + traffic passes of the default command, the two reset
# modes (bench lines + kernel traces), in-kernel stamps.   bash tools/round_evidence.sh r03   ->   gpurun_out/<tag>_*
set -e
TAG=$1
O=gpurun_out
bash tools/profile_round.sh $TAG
timeout -k 10 280 python3 bench.py --no-cpu-baseline --auto-reset masked > $O/${TAG}_bench_auto_reset_masked.json 2> $O/${TAG}_masked.err
timeout -k 10 280 python3 bench.py --no-cpu-baseline --auto-reset masked --events > $O/${TAG}_bench_auto_reset_masked_events.json 2> $O/${TAG}_events.err
timeout -k 10 280 python3 bench.py --no-cpu-baseline --normalize-obs > $O/${TAG}_bench_normalize_obs.json 2> $O/${TAG}_norm.err
bash tools/profile_modes.sh $TAG
{
  echo "# in-kernel stamps (make STAMPS=1; tools/kernel_stamps.py), 4096 envs, Anymal-C rough"
  echo "## post-physics launch without resets, actuator launches"
  MDP_HIP_LIB=isaaclab_amd/_lib/libmdp_hip_stamps.so timeout -k 10 200 python3 tools/kernel_stamps.py 2>&1 | grep -v "^$" | grep -v amdgpu.ids
  echo "## post-physics launch with the reset branch (masked resets)"
  STAMPS_RESET=1 MDP_HIP_LIB=isaaclab_amd/_lib/libmdp_hip_stamps.so timeout -k 10 200 python3 tools/kernel_stamps.py 2>&1 | grep -v "^$" | grep -v amdgpu.ids | head -14
  echo "## ... with curriculum and event terms folded in"
  STAMPS_RESET=1 STAMPS_EVENTS=1 MDP_HIP_LIB=isaaclab_amd/_lib/libmdp_hip_stamps.so timeout -k 10 200 python3 tools/kernel_stamps.py 2>&1 | grep -v "^$" | grep -v amdgpu.ids | head -14
} > $O/${TAG}_stamps.txt
tail -5 $O/${TAG}_stamps.txt | cut -c1-200
