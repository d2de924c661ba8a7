#!/bin/bash
# GPU box: rocprofv3 kernel trace of the multi-model loop (tools/profile_frames.py): per-kernel statistics + the timeline of one frame
#   tools/mm_trace.sh <models> <tag> ["ENV=.."]
m=$1; tag=$2; cfg=${3:-MMF_NOP=1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mmt_$tag -o p -- python3 $R/tools/profile_frames.py 150 640x480 $m 1 > $R/gpurun_out/mmt_$tag.log 2>&1
grep -h 'ms per frame' $R/gpurun_out/mmt_$tag.log
csv=$(find $R/gpurun_out/mmt_$tag -name '*kernel_trace.csv' | head -1)
python3 $R/tools/kstats_csv.py $csv 150 > $R/gpurun_out/mmt_${tag}_stats.txt
python3 $R/tools/timeline.py $csv 100 > $R/gpurun_out/mmt_${tag}_timeline.txt
rm -rf $R/gpurun_out/mmt_$tag
head -30 $R/gpurun_out/mmt_${tag}_stats.txt
