#!/bin/bash
# experiment (GPU box): kernel durations of steady-state frames WITHOUT the prefetch overlap, for the shipped library and for
# variant libraries built into ab/ (selected through MMF_HIP_LIB; the shipped library is never touched).
#   bash tools/variant_prof.sh "<pattern>" shipped v1 v2 ...
R=$GRAFT_REPO_ROOT
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = shipped ]; then unset MMF_HIP_LIB; else export MMF_HIP_LIB=$R/ab/libmmf_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vp_$v -o p -- python3 $R/tools/profile_frames.py ${FRAMES:-100} ${SIZE:-640x480} ${MODELS:-1} 0 > $R/gpurun_out/vp_$v.log 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/vp_$v/p_kernel_stats.csv")))
print("== $v", open("$R/gpurun_out/vp_$v.log").read().strip().splitlines()[-1])
for r in rows:
    if any(p in r["Name"] for p in "$pat".split("|")):
        print("  %-46s n=%5s avg %7.2f min %7.2f us"%(r["Name"].replace("mmf::","")[:46], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
