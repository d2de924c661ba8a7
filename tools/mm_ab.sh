#!/bin/bash
# GPU box: ms per frame with several models on the GPU (tools/profile_frames.py, prefetch on) under several environments
#   tools/mm_ab.sh <models> [reps] "ENV=.." ...      ("-" = no extra environment)
m=$1; reps=$2; shift; shift
for r in $(seq 1 $reps); do
  for cfg in "$@"; do
    c="$cfg"; [ "$c" = "-" ] && c="MMF_NOP=1"
    echo "$m models, $cfg: $(env $c timeout -k 10 200 python tools/profile_frames.py 140 640x480 $m 1 2>/dev/null | grep 'ms per frame')"
  done
done
