cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r2b_t4.log 2>&1 || { tail -40 gpurun_out/r2b_t4.log; exit 1; }
tail -3 gpurun_out/r2b_t4.log
bash tools/measure_configs.sh r02b > gpurun_out/r02b_measure.log 2>&1; tail -30 gpurun_out/r02b_measure.log | cut -c1-420
