#!/bin/bash
# Round 6: the one-window encoder GEMM (gemm2_kernel) with full-128-byte-line swizzled activation pieces — A/B against the fragment-shaped pieces
# (whisperlive_amd/libwlx_fragb.so = the same tree with the previous gemm.hip), then the encoder parity / bit-identity tests and rocprof of the encoder.
set -u
TAG=${1:-r6n}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for m in small.en large-v3 tiny.en; do for lib in libwlx.so libwlx_fragb.so libwlx.so libwlx_fragb.so; do
  echo "== $m $lib"; WLX_LIB=whisperlive_amd/$lib timeout 300 python scripts/encode_only.py $m 20 1 2>/dev/null | tail -1
done; done 2>&1 | tee "$OUT/gemm2_fullline_ab.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_encoder_batched.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -4 | tee "$OUT/pytest_tail.txt"
cd /tmp
for lib in libwlx.so libwlx_fragb.so; do
  WLX_LIB=$GRAFT_REPO_ROOT/whisperlive_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$lib" -o enc -- python $GRAFT_REPO_ROOT/scripts/encode_only.py small.en 10 1 > /dev/null 2>&1
  f=$(find "$OUT/prof_$lib" -name '*kernel_stats.csv' | head -1); echo "== $lib"; head -12 "$f" | cut -c1-200
done 2>&1 | tee "$OUT/kernel_stats_ab.txt"
