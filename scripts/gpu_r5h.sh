#!/bin/bash
# Rows per row tile at >= 120 rows (two / three MFMA row tiles per workgroup share a pass over the weight tile): step time by model and rows.
set -u
TAG=${1:-r5h}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=$REPO/whisperlive_amd/libwlx_ab.so
for cfg in "small.en 120" "small.en 240" "large-v3 80" "large-v3 160"; do
  set -- $cfg
  for ch in 16 32 48; do
    WLX_ROWTILE_CHUNK=$ch timeout 300 python scripts/step_profile.py $1 $2 33 2>/dev/null | head -12 | tee -a "$OUT/rowtile_chunk_$1_$2.txt" | head -8
  done
done
