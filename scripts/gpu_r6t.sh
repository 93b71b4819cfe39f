#!/bin/bash
# Round 6: encoder attention launch time against workgroups per CU (tiny 144, base 192, small 288, medium 384, large-v3 480 workgroups of 4 waves on
# 256 CUs): is the launch as long as its busiest SIMD's wave count?
set -u
TAG=${1:-r6t}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
REPO=$PWD; cd /tmp
for m in tiny.en base.en small.en medium.en large-v3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$m" -o enc --output-format csv -- python $REPO/scripts/encode_only.py $m 6 1 > /dev/null 2>&1
  f=$(find "$OUT/prof_$m" -name '*kernel_stats.csv' | head -1); echo "== $m"; grep -E "attn_encoder|layernorm_kernelILb0|gemm2" "$f" | cut -d, -f1-4,6,7 | cut -c1-200
  rm -rf "$OUT/prof_$m"
done 2>&1 | tee "$OUT/attention_by_model.txt"
