# Same-box A/B of library variants on the whole bench step: bash tools/ab_bench.sh "main nt main nt" [bench args]
V=$1; shift
for v in $V; do
  lib=isaaclab_amd/_lib/libmdp_hip_$v.so; [ "$v" = main ] && lib=isaaclab_amd/_lib/libmdp_hip.so
  echo "== $v"
  MDP_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --regions 30 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%.2f M env-steps/s  %.2f us/step  '%(d['value']/1e6, d['ms_per_step']*1e3) + '  '.join('%s %.2f us'%(k,v['us']) for k,v in r['kernels'].items()))" || exit 1
done
