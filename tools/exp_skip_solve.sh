#!/bin/bash
# experiment (GPU box): how much of rgb_step_kernel is the single-lane solve?  The variant library is built to /tmp and
# selected through MMF_HIP_LIB: the shipped multimotionfusion_amd/libmmf_hip.so is never touched.
set -e
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -w -DMMF_SKIP_SOLVE -o /tmp/libmmf_skip.so $R/multimotionfusion_amd/csrc/mmf_hip.hip
cd /tmp && export TMPDIR=/tmp
MMF_HIP_LIB=/tmp/libmmf_skip.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_skip -o skip -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_skip.log 2>&1 || true
