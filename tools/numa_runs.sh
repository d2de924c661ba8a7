#!/bin/bash
# GPU box: the headline loop pinned to the cores next to the GPU and to the cores far from it
lscpu | grep -E "^CPU\(s\)|NUMA|Socket|Model name" 
for d in /sys/class/drm/card*/device; do
  if [ -f $d/vendor ] && grep -q 0x1002 $d/vendor; then echo "$d numa_node $(cat $d/numa_node) local_cpulist $(cat $d/local_cpulist)"; fi
done
nproc; taskset -p $$
python - <<'PY'
import os
print("affinity of this process:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], "...")
PY
run() {
  for r in 1 2 3; do
    MMF_HOST_TRACE=1 MMF_BENCH_HEADLINE_ONLY=1 timeout -k 10 200 taskset -c $1 python bench.py --steps 600 --warmup 30 --no-cpu-baseline --no-extras 2> gpurun_out/numa.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['gn_chain']['per_level']
print('cpus $1: %.0f fps  chain %.1f us  l0 %.2f l1 %.2f l2 %.2f us' % (d['value'], d['gn_chain']['us'], p['l0']['producer_us'], p['l1']['producer_us'], p['l2']['producer_us']))"
    grep -h "host us" gpurun_out/numa.err | tail -1 | cut -c1-80
  done
}
for set in "$@"; do run $set; done
