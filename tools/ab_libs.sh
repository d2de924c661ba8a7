#!/bin/bash
# GPU box: same-box A/B of library builds on the headline loop (frames/s, chain us, per-level launch us).
#   tools/ab_libs.sh [reps] build/libmmf_a.so build/libmmf_b.so ...
reps=$1; shift
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    MMF_HIP_LIB=$PWD/$lib MMF_BENCH_HEADLINE_ONLY=1 timeout -k 10 120 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['gn_chain']['per_level']
    print('$lib: %.0f fps  chain %.1f us  l0 %.2f l1 %.2f l2 %.2f us' % (d['value'], d['gn_chain']['us'], p['l0']['producer_us'], p['l1']['producer_us'], p['l2']['producer_us']))
except Exception as e:
    print('$lib: failed', e)"
  done
done
