# Copies what tools/round_evidence.sh <src-tag> (and tools/measure_configs.sh <src-tag>, if it ran) left under gpurun_out/
# into profiles/ under the round's names:   bash tools/publish_evidence.sh r03c r03
set -e
S=$1; T=$2; O=gpurun_out; P=profiles
C=$(git rev-parse --short HEAD)
last() { tail -n 1 "$1"; }
last $O/${S}_bench.json > $P/${T}_bench.json
for m in auto_reset_masked auto_reset_masked_events normalize_obs; do
  [ -s $O/${S}_bench_$m.json ] && last $O/${S}_bench_$m.json > $P/${T}_bench_$m.json
done
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --regions 20   (round ${T#r}, tree $C; frame ring of 4 graph segments replayed in rotation: cold frames)"; cat $O/${S}_kernel_trace.txt; } > $P/${T}_bench_graph_kernel_trace.txt
[ -s $O/${S}_kernel_trace_masked.txt ] && { echo "# the same with --auto-reset masked"; cat $O/${S}_kernel_trace_masked.txt; } > $P/${T}_kernel_trace_auto_reset_masked.txt
[ -s $O/${S}_kernel_trace_events.txt ] && { echo "# the same with --auto-reset masked --events"; cat $O/${S}_kernel_trace_events.txt; } > $P/${T}_kernel_trace_auto_reset_masked_events.txt
python3 - $O/${S}_pmc_traffic.json $P/${T}_pmc_traffic.json $P/pmc_traffic.json $T $C <<'PY'
import json, sys
src, out1, out2, tag, commit = sys.argv[1:]
d = json.load(open(src))
d["source"] = {"tag": tag, "commit": commit, "script": "tools/profile_round.sh",
               "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --regions 3"}
for o in (out1, out2):
    json.dump(d, open(o, "w"), indent=1)
PY
cp $O/${S}_stamps.txt $P/${T}_stamps.txt
for f in C1_anymal_d_flat_4096 C2_anymal_d_rough_4096 C3_anymal_c_rough_16384 C4_h1_rough_8192 C5_shadow_16384 anymal_c_rough_65536; do
  [ -s $O/${S}_$f.json ] && last $O/${S}_$f.json > $P/${T}_bench_$f.json
done
[ -s $O/${S}_pmc_mfma.txt ] && cp $O/${S}_pmc_mfma.txt $P/${T}_pmc_mfma_lstm_16384.txt
echo published $S as $T at $C
