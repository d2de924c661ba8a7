#!/bin/bash
# GPU box: derived counters of the kernels matching a pattern, one rocprofv3 --pmc pass per counter group, mean over launches
#   tools/pmc_kernel.sh <pattern> "<counters of pass 1>" "<counters of pass 2>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; pat=$1; shift; i=0
for grp in "$@"; do
  i=$((i+1)); O=$R/gpurun_out/pmck$i; rm -rf $O
  rocprofv3 --pmc $grp -d $O -o p --output-format csv -- python3 $R/tools/profile_frames.py 150 640x480 1 0 headline > $O.log 2>&1
  python3 - "$O" "$pat" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if sys.argv[2] in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    v = v[len(v) // 2:]  # the later launches: the map has grown
    print("%-28s mean %14.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
done
