#!/bin/bash
# Round 4: residual epilogue batch shapes of the large-M form (n-tiles x m-tiles requested together), one box, interleaved
set -u
TAG=${1:-r4resid4}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
enc() { env $1 timeout 300 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s|^|[$1] |" | tee -a "$OUT/encode_ab.txt"; }
L=$PWD/whisperlive_amd
for rep in 1 2; do
  enc A=tile_by_tile small.en 12
  enc WLX_LIB=$L/libwlx_r41.so small.en 12
  enc WLX_LIB=$L/libwlx_r42.so small.en 12
  enc WLX_LIB=$L/libwlx_r12.so small.en 12
done
enc A=tile_by_tile large-v3 8
enc WLX_LIB=$L/libwlx_r41.so large-v3 8
enc WLX_LIB=$L/libwlx_r42.so large-v3 8
echo done
