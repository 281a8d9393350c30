# kernel-trace summaries of the reset modes (GPU box): bash tools/profile_modes.sh r03
set -e
TAG=$1
O=gpurun_out
mkdir -p $O
for mode in masked events; do
  args="--auto-reset masked"; [ $mode = events ] && args="--auto-reset masked --events"
  rm -rf $O/${TAG}_kt_$mode
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt_$mode -o kt -- python3 bench.py --no-cpu-baseline --regions 20 $args > $O/${TAG}_kt_$mode.log 2>&1
  python3 tools/rocprof_summary.py $(find $O/${TAG}_kt_$mode -name "*_results.db" | head -1) > $O/${TAG}_kernel_trace_$mode.txt 2>&1 || true
  find $O/${TAG}_kt_$mode -name "*.db" -delete
  tail -1 $O/${TAG}_kt_$mode.log | head -c 400; echo
  head -30 $O/${TAG}_kernel_trace_$mode.txt
done
