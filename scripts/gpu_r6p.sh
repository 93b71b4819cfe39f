#!/bin/bash
# Round 6: gemm2 tile shape per encoder GEMM after the full-line activation pieces — each of the three shapes forced for every launch
# (libwlx_ab.so, WLX_GEMM2_SHAPE=0/1/2) and the production pick, per (kernel, grid) launch time from rocprofv3 --kernel-trace.
set -u
TAG=${1:-r6p}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
REPO=$PWD; cd /tmp
for m in small.en large-v3; do for sh in pick 0 1 2; do
  if [ $sh = pick ]; then unset WLX_GEMM2_SHAPE; else export WLX_GEMM2_SHAPE=$sh; fi
  WLX_LIB=$REPO/whisperlive_amd/libwlx_ab.so timeout 300 rocprofv3 --kernel-trace -d "$OUT/prof_${m}_$sh" -o enc --output-format csv -- python $REPO/scripts/encode_only.py $m 6 1 2>/dev/null | tail -1
  python $REPO/scripts/trace_by_grid.py "$OUT/prof_${m}_$sh" "$m shape=$sh" 6
  rm -rf "$OUT/prof_${m}_$sh"
done; done 2>&1 | tee "$OUT/gemm2_shapes_by_grid.txt"
