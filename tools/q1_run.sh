#!/bin/bash
# GPU box: kernel time on the model's stream per frame (tools/q1_sum.py) of the headline loop under several environments
#   tools/q1_run.sh "A=1" "MMF_X=0" ...   -> gpurun_out/q1_<n>.txt
cd /tmp && export TMPDIR=/tmp
n=0
for cfg in "$@"; do
  n=$((n+1))
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_q
  export $cfg
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q -o p -- python3 $GRAFT_REPO_ROOT/tools/profile_frames.py 120 640x480 1 ${MMF_Q1_HINT:-1} headline > $GRAFT_REPO_ROOT/gpurun_out/prof_q.log 2>&1
  unset ${cfg%%=*}
  echo "== $cfg" > $GRAFT_REPO_ROOT/gpurun_out/q1_$n.txt
  python3 $GRAFT_REPO_ROOT/tools/q1_sum.py $(ls $GRAFT_REPO_ROOT/gpurun_out/prof_q/*kernel_trace.csv | head -1) >> $GRAFT_REPO_ROOT/gpurun_out/q1_$n.txt
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/q1_$n.txt
done
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_q
