#!/bin/bash
# GPU box: ms per frame with several models on the GPU for several library builds, alternating
#   tools/mm_ab_libs.sh <models> <reps> build/libmmf_a.so ...
m=$1; reps=$2; shift; shift
for r in $(seq 1 $reps); do
  for lib in "$@"; do
    echo "$m models, $lib: $(MMF_HIP_LIB=$PWD/$lib timeout -k 10 200 python tools/profile_frames.py 140 640x480 $m 1 2>/dev/null | grep 'ms per frame')"
  done
done
