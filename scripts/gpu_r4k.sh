#!/bin/bash
# round 4, call K: the 4-wave LDS-DMA-ring GEMM (gemm2) at large M by tile shape, incl. a 2-workgroups-per-CU 128 x 128 form
set -u
TAG=${1:-r4k}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
enc() { env $1 timeout 600 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s/^/$1 /"; }
{
enc "WLX_GEMM3=1" small.en 12
for SH in 0 1 2 3 4 5 7; do enc "WLX_GEMM3=0 WLX_GEMM2_SHAPE=$SH" small.en 12; done
enc "WLX_GEMM3=1" large-v3 8
for SH in 0 7; do enc "WLX_GEMM3=0 WLX_GEMM2_SHAPE=$SH" large-v3 8; done
for B in 2 3; do enc "WLX_GEMM3=2" small.en $B; enc "WLX_GEMM3=0" small.en $B; done
} | tee "$OUT/encode_shape_times.txt"
