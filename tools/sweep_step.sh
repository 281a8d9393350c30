# whole-step sweep of env-var knobs (GPU box): bash tools/sweep_step.sh "MDP_POST_ENVS=16 MDP_POST_ENVS=24 ..." [bench args]
K=$1; shift
for kv in $K; do
  echo "== $kv"
  env $(echo $kv | tr ',' ' ') timeout -k 10 200 python3 bench.py --no-cpu-baseline --regions 30 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%.2f M env-steps/s  %.2f us/step  '%(d['value']/1e6, d['ms_per_step']*1e3) + '  '.join('%s %.2f us'%(k,v['us']) for k,v in r['kernels'].items()))" || exit 1
done
