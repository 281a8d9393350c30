# second half of tools/measure_configs.sh (C4, C5, large N, MFMA counter pass)
set -e
TAG=$1; shift
O=gpurun_out
mkdir -p $O
B="timeout -k 10 280 python3 bench.py --no-cpu-baseline"
$B --config h1_rough                      > $O/${TAG}_C4_h1_rough_8192.json        2> $O/${TAG}_C4.err
echo C4 done
timeout -k 10 280 python3 bench.py --config shadow --cpu-steps 16 > $O/${TAG}_C5_shadow_16384.json 2> $O/${TAG}_C5.err
echo C5 done
$B --config anymal_c_rough --envs 65536 --ring 8 > $O/${TAG}_anymal_c_rough_65536.json 2> $O/${TAG}_N65536.err
echo N65536 done
rocprofv3 -L > $O/${TAG}_counters.txt 2>&1 || true
CNT=""
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU; do
  if grep -qw $c $O/${TAG}_counters.txt; then CNT="$CNT $c"; fi
done
echo "mfma pass counters:$CNT"
rm -rf $O/${TAG}_pmcM
timeout -s KILL 280 rocprofv3 --pmc $CNT GRBM_GUI_ACTIVE -d $O/${TAG}_pmcM -o m -- python3 bench.py --no-cpu-baseline --regions 3 --envs 16384 > $O/${TAG}_pmcM.log 2>&1
python3 tools/rocprof_pmc.py $(find $O/${TAG}_pmcM -name "*_results.db" | head -1) > $O/${TAG}_pmc_mfma.txt 2>&1 || true
find $O/${TAG}_pmcM -name "*.db" -delete
grep -E "k_lstm|k_post" $O/${TAG}_pmc_mfma.txt || true
for f in $O/${TAG}_C4*.json $O/${TAG}_C5*.json $O/${TAG}_anymal_c_rough_65536.json; do echo $f; head -c 300 $f; echo; done
