#!/bin/bash
# as tools/ab_libs.sh, 600 timed steps (the map of the headline loop reaches its steady size) and the surfel passes' times
reps=$1; shift
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    MMF_HIP_LIB=$PWD/$lib MMF_BENCH_HEADLINE_ONLY=1 timeout -k 10 200 python bench.py --steps 600 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib: %.0f fps  surfels %d  combinedPredict %.1f us  predictIndices %.1f us  chain %.1f us' % (d['value'], d['surfels'], d['surfel_passes']['combinedPredict']['us'], d['surfel_passes']['predictIndices']['us'], d['gn_chain']['us']))"
  done
done
