#!/bin/bash
# GPU box: splat_kernel's time with parts of it compiled out (build/libmmf_sabl<N>.so = -DMMF_SPLAT_ABL=N), kernel trace
#   tools/splat_abl.sh <layers> N...
cd /tmp && export TMPDIR=/tmp
L=$1; shift
for a in "$@"; do
  export MMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libmmf_sabl$a.so
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/sabl$a
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sabl$a -- python3 $GRAFT_REPO_ROOT/tools/mature_splat_probe.py 640x480 $L > $GRAFT_REPO_ROOT/gpurun_out/sabl$a.log 2>&1
  echo "ABL $a layers $L: $(python3 $GRAFT_REPO_ROOT/tools/kmedian.py $GRAFT_REPO_ROOT/gpurun_out/sabl$a splat_kernel splat_resolve_fill 2>&1 | tail -1)"
done
