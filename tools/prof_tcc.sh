# TCC counters of the cast / post-physics launches: bash tools/prof_tcc.sh <tag>   (env passes through)
set -e
P=gpurun_out/prof_tcc_$1
rm -rf $P; mkdir -p $P
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $P/p -o p -- python3 tools/prof_cast.py > $P/p.log 2>&1
python3 tools/rocprof_pmc.py $(find $P/p -name "*_results.db" | head -1) 2>&1 | grep -E "k_raycaster_update_down|k_post_physics"
