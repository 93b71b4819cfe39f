#!/bin/bash
# Round 4: tile shape of the second form for the batched conv front end (zbatch = windows): WLX_GEMM2_LARGE_SHAPE
set -u
TAG=${1:-r4conv}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
enc() { env $1 timeout 300 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s|^|[$1] |" | tee -a "$OUT/encode_ab.txt"; }
for rep in 1 2; do for sh in 3 0 1 2 7; do enc WLX_GEMM2_LARGE_SHAPE=$sh small.en 12; done; done
for sh in 3 0 7; do enc WLX_GEMM2_LARGE_SHAPE=$sh large-v3 8; done
echo done
