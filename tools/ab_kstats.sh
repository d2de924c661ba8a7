#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the headline loop under several environments, same box, one after the other.
#   tools/ab_kstats.sh "grep-pattern" "ENV=.." "ENV=.." ...
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg rocprofv3 --kernel-trace --stats -d $R/gpurun_out/abk_$i -o p -- python3 $R/tools/profile_frames.py 150 640x480 1 1 headline > $R/gpurun_out/abk_$i.log 2>&1
  echo "== $cfg: $(grep -h 'frames/s\|ms per frame' $R/gpurun_out/abk_$i.log | tail -1)"
  python3 $R/tools/kstats.py $(ls $R/gpurun_out/abk_$i/*results.db | head -1) 150 | grep -i "per frame\|$pat"
  rm -rf $R/gpurun_out/abk_$i
done
