#!/bin/bash
# as tools/ab_env.sh, 600 timed steps (the map of the headline loop reaches its steady size)
reps=$1; shift
for r in $(seq 1 $reps); do
  for cfg in "$@"; do
    c="$cfg"; [ "$c" = "-" ] && c="MMF_NOP=1"
    env $c MMF_BENCH_HEADLINE_ONLY=1 timeout -k 10 200 python bench.py --steps 600 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg: %.0f fps  surfels %d  combinedPredict %.1f us  predictIndices %.1f us' % (d['value'], d['surfels'], d['surfel_passes']['combinedPredict']['us'], d['surfel_passes']['predictIndices']['us']))"
  done
done
