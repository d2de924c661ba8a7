#!/bin/bash
# experiment (GPU box): surfel-pass kernel durations on a fixed map for the shipped library and variants in ab/
#   bash tools/probe_variants.sh "<pattern>" shipped v1 ...     (SIZE=1280x960 for the large frame)
R=$GRAFT_REPO_ROOT
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/surfel_probe.py build ${SIZE:-640x480} ${FRAMES:-30} || exit 1
for v in "$@"; do
  if [ $v = shipped ]; then unset MMF_HIP_LIB; else export MMF_HIP_LIB=$R/ab/libmmf_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sp_$v -o p -- python3 $R/tools/surfel_probe.py run ${ROUNDS:-30} > $R/gpurun_out/sp_$v.log 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/sp_$v/p_kernel_stats.csv")))
print("== $v", [l for l in open("$R/gpurun_out/sp_$v.log").read().splitlines() if l.startswith("rounds")])
for r in rows:
    if any(p in r["Name"] for p in "$pat".split("|")):
        print("  %-46s n=%5s avg %7.2f min %7.2f us"%(r["Name"].replace("mmf::","")[:46], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
