#!/bin/bash
# experiment (GPU box): alternate bench.py between the shipped library and ab/libmmf_<name>.so in one process
# sequence on one box (box-to-box and run-to-run spread is larger than most effects; compare the best of each)
#   bash tools/ab_bench.sh <name> [rounds] [steps]
R=$GRAFT_REPO_ROOT
cd $R
for r in $(seq 1 ${2:-3}); do
  for v in shipped $1; do
    if [ $v = shipped ]; then unset MMF_HIP_LIB; else export MMF_HIP_LIB=$R/ab/libmmf_$v.so; fi
    python bench.py --no-cpu-baseline --steps ${3:-300} > gpurun_out/abb_$v.json 2>> gpurun_out/abb.err
    python - <<PY
import json
r=json.loads(open("gpurun_out/abb_$v.json").read().strip().splitlines()[-1])
print("%-8s %6.0f frames/s  %.4f ms"%("$v", r["value"], r["ms_per_step"]))
PY
  done
done
