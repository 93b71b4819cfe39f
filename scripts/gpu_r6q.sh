#!/bin/bash
# Round 6: the re-fitted gemm2 shape pick against each shape forced (libwlx_ab.so), encoder time of one window, every model size.
set -u
TAG=${1:-r6q}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for m in tiny.en base.en small.en medium.en large-v3; do for sh in pick 0 1 2 pick; do
  if [ $sh = pick ]; then unset WLX_GEMM2_SHAPE; else export WLX_GEMM2_SHAPE=$sh; fi
  echo -n "$m shape=$sh  "; WLX_LIB=whisperlive_amd/libwlx_ab.so timeout 300 python scripts/encode_only.py $m 20 1 2>/dev/null | tail -1
done; done 2>&1 | tee "$OUT/gemm2_pick_vs_forced.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_encoder_batched.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -3 | tee "$OUT/pytest_tail.txt"
