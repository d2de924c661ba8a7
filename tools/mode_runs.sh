#!/bin/bash
# GPU box: the headline loop several times in a row with the calling thread's time line (MMF_HOST_TRACE=1): which runs land in
# the slow mode of the level-0 launch, and where the host is at that time; alternating between environments
#   tools/mode_runs.sh <runs> "ENV=.." ...     ("-" = no extra environment)
n=$1; shift
for r in $(seq 1 $n); do
  for cfg in "$@"; do
    c="$cfg"; [ "$c" = "-" ] && c="MMF_NOP=1"
    env $c MMF_HOST_TRACE=1 MMF_BENCH_HEADLINE_ONLY=1 timeout -k 10 200 python bench.py --steps 600 --warmup 30 --no-cpu-baseline --no-extras 2> gpurun_out/mode.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['gn_chain']['per_level']
print('$cfg: %.0f fps  chain %.1f us  l0 %.2f l1 %.2f l2 %.2f us' % (d['value'], d['gn_chain']['us'], p['l0']['producer_us'], p['l1']['producer_us'], p['l2']['producer_us']), end='  ')"
    grep -h "host us" gpurun_out/mode.err | head -1 | cut -c25-75
  done
done
