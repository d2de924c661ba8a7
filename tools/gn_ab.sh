#!/bin/bash
# GPU box: A/B of environment knobs on the headline loop; prints frames/s, chain us and the per-level launch times.
#   tools/gn_ab.sh "MMF_GN_SLEEP=1" "MMF_GN_SLEEP=8" ...
for cfg in "$@"; do
  env $cfg MMF_BENCH_HEADLINE_ONLY=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['gn_chain']['per_level']
print('$cfg: %.0f fps  chain %.1f us  l0 %.2f l1 %.2f l2 %.2f us' % (d['value'], d['gn_chain']['us'], p['l0']['producer_us'], p['l1']['producer_us'], p['l2']['producer_us']))"
done
