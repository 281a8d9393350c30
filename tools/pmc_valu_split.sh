# Vector-instruction count of the post-physics launch split by role (GPU box): one SQ pass over tools/prof_cast.py, which
# runs the stand-alone vertical-ray kernel (= the cast role alone) and the post-physics launch on the same frames.
#   bash tools/pmc_valu_split.sh   ->   gpurun_out/valu_split.txt
P=gpurun_out/valu_split
rm -rf $P; mkdir -p $P
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $P/p -o p -- python3 tools/prof_cast.py > $P/p.log 2>&1
python3 tools/rocprof_pmc.py $(find $P/p -name "*_results.db" | head -1) 2>&1 | grep -E "k_raycaster_update_down|k_post_physics|k_lstm8" | cut -c1-420 > gpurun_out/valu_split.txt
find $P -name "*.db" -delete
cat gpurun_out/valu_split.txt
