#!/bin/bash
# GPU box: same-box A/B of environment settings (and library builds: MMF_HIP_LIB=...) on the headline loop.
#   tools/ab_env.sh [reps] "ENV=.. ENV=.." "ENV=.." ...      ("-" = no extra environment)
reps=$1; shift
for r in $(seq 1 $reps); do
  for cfg in "$@"; do
    c="$cfg"; [ "$c" = "-" ] && c="MMF_NOP=1"
    env $c MMF_BENCH_HEADLINE_ONLY=1 timeout -k 10 120 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['gn_chain']['per_level']
    print('$cfg: %.0f fps  chain %.1f us  l0 %.2f l1 %.2f l2 %.2f us' % (d['value'], d['gn_chain']['us'], p['l0']['producer_us'], p['l1']['producer_us'], p['l2']['producer_us']))
except Exception as e:
    print('$cfg: failed', e)"
  done
done
