#!/bin/bash
# GPU box: ms per frame of the several-models-per-GPU workload under several environments (same box, one after the other)
#   tools/mm_ab.sh <models> "ENV=.." "ENV=.." ...
m=$1; shift
for cfg in "$@"; do
  echo "$cfg: $(env $cfg python tools/profile_frames.py 160 640x480 $m 1 2>/dev/null | grep 'ms per frame')"
done
