O=gpurun_out
rm -rf $O/kt_shadow
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt_shadow -o kt -- python3 bench.py --no-cpu-baseline --regions 10 --config shadow > $O/kt_shadow.log 2>&1
python3 tools/rocprof_summary.py $(find $O/kt_shadow -name "*_results.db" | head -1) 2>&1 | head -40 | cut -c1-170
find $O/kt_shadow -name "*.db" -delete
