#!/bin/bash
# experiment (GPU box): how much of rgb_step_kernel is the single-lane solve?
set -e
R=$GRAFT_REPO_ROOT
cp $R/multimotionfusion_amd/libmmf_hip.so /tmp/lib_orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -w -DMMF_SKIP_SOLVE -o $R/multimotionfusion_amd/libmmf_hip.so $R/multimotionfusion_amd/csrc/mmf_hip.hip
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_skip -o skip -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_skip.log 2>&1 || true
cp /tmp/lib_orig.so $R/multimotionfusion_amd/libmmf_hip.so
