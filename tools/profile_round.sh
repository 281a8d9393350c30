# Round profile of the default bench command (GPU box): bench line, rocprofv3 kernel-trace summary, FETCH/WRITE passes.
#   bash tools/profile_round.sh r02 [extra bench args]   ->   gpurun_out/<tag>_*   (copy what is judged into profiles/)
set -e
TAG=$1; shift
O=gpurun_out
mkdir -p $O
timeout -k 10 300 python3 bench.py "$@" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rm -rf $O/${TAG}_kt $O/${TAG}_pmcF $O/${TAG}_pmcW
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt -o kt -- python3 bench.py --no-cpu-baseline --regions 20 "$@" > $O/${TAG}_kt.log 2>&1
python3 tools/rocprof_summary.py $(find $O/${TAG}_kt -name "*_results.db" | head -1) > $O/${TAG}_kernel_trace.txt 2>&1 || true
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_pmcF -o f -- python3 bench.py --no-cpu-baseline --regions 3 "$@" > $O/${TAG}_pmcF.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_pmcW -o w -- python3 bench.py --no-cpu-baseline --regions 3 "$@" > $O/${TAG}_pmcW.log 2>&1
python3 tools/pmc_traffic.py $(find $O/${TAG}_pmcF -name "*_results.db" | head -1) $(find $O/${TAG}_pmcW -name "*_results.db" | head -1) -o $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt 2>&1 || true
find $O/${TAG}_kt $O/${TAG}_pmcF $O/${TAG}_pmcW -name "*.db" -delete
head -c 600 $O/${TAG}_bench.json; echo; head -12 $O/${TAG}_kernel_trace.txt; grep -E "k_post_physics|k_lstm" $O/${TAG}_pmc_traffic.txt
