#!/bin/bash
# round 3, call J: self-attention waves per (row, head): 1 / 4 / 8 by position
set -u
TAG=r3j; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for lib in libwlx_sa1.so libwlx.so libwlx_sa8.so; do
  WLX_LIB=whisperlive_amd/$lib timeout 300 python scripts/step_by_position.py small.en 2>&1 | grep "decode step"
  WLX_LIB=whisperlive_amd/$lib timeout 300 python scripts/step_by_position.py large-v3 2>&1 | grep "decode step"
done > "$OUT/step_by_position.txt" 2>&1; cat "$OUT/step_by_position.txt"
