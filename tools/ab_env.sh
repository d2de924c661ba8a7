#!/bin/bash
# experiment (GPU box): alternate bench.py with and without an environment switch in one box
#   bash tools/ab_env.sh MMF_GRAPHS=1 [rounds] [steps]
R=$GRAFT_REPO_ROOT
cd $R
for r in $(seq 1 ${2:-3}); do
  for v in default "$1"; do
    if [ "$v" = default ]; then out=$(python bench.py --no-cpu-baseline --steps ${3:-300} 2>> gpurun_out/abe.err); else out=$(env $1 python bench.py --no-cpu-baseline --steps ${3:-300} 2>> gpurun_out/abe.err); fi
    echo "$out" | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s %6.0f frames/s  %.4f ms  chain %.1f us  mm %s'%('$v', r['value'], r['ms_per_step'], r['gn_chain']['us'], [round(x['model_frames_per_s']) for x in r.get('multi_model',{}).get('sweep',[])] if isinstance(r.get('multi_model'),dict) else ''))"
  done
done
