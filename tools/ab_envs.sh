#!/bin/bash
# experiment (GPU box): bench.py under several environment settings, alternating, in one box
#   bash tools/ab_envs.sh <rounds> <steps> "A=1" "B=1 C=2" ...     ("-" = no extra setting)
R=$GRAFT_REPO_ROOT
cd $R
rounds=$1; steps=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then out=$(MMF_BENCH_SKIP_CONFIG5=1 python bench.py --no-cpu-baseline --steps $steps 2>> gpurun_out/abe.err); else out=$(env MMF_BENCH_SKIP_CONFIG5=1 $v python bench.py --no-cpu-baseline --steps $steps 2>> gpurun_out/abe.err); fi
    echo "$out" | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %6.0f frames/s  %.4f ms  chain %.1f us'%('$v', r['value'], r['ms_per_step'], r['gn_chain']['us']))"
  done
done
