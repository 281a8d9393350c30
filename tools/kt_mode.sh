# rocprofv3 kernel-trace summary of a bench mode (GPU box): bash tools/kt_mode.sh <tag> [bench args]
O=gpurun_out; T=$1; shift
rm -rf $O/kt_$T
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt_$T -o kt -- python3 bench.py --no-cpu-baseline --regions 10 "$@" > $O/kt_$T.log 2>&1
python3 tools/rocprof_summary.py $(find $O/kt_$T -name "*_results.db" | head -1) 2>&1 | head -40 | cut -c1-150 | tee $O/kt_$T.txt
find $O/kt_$T -name "*.db" -delete
