#!/bin/bash
# GPU box: the measured FLOOR of a gn_iter_kernel launch (csrc/gn_fused.hpp): diagnostic builds of the library whose launches
# keep every dependent phase -- previous sums -> (solve) -> pose broadcast -> arrival -> count barrier -> fixed-point sums --
# and do NO pixel work (-DMMF_ABL=64), and the same without the solve (-DMMF_ABL=4160), timed by the bench's own per-launch
# HIP events beside the shipped library.  What the shipped launch takes above the first figure is its pixel work.
#   tools/gn_floor_probe.sh > profiles/r04_gn_floor_probe.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/build
for v in 64 4160; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -I/opt/rocm/include -DMMF_ABL=$v \
      -o $R/build/libmmf_floor$v.so $R/multimotionfusion_amd/csrc/mmf_hip.hip -ldl || exit 1
done
cd $R
echo "per-launch HIP events of the headline loop (us): shipped library, then no pixel work, then no pixel work and no solve"
echo "(the diagnostic builds track nothing: their frames/s and chain figures only say the loop ran)"
tools/ab_libs.sh 2 multimotionfusion_amd/libmmf_hip.so build/libmmf_floor64.so build/libmmf_floor4160.so
rm -f build/libmmf_floor64.so build/libmmf_floor4160.so
