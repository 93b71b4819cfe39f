#!/bin/bash
# Round 6: the single-window cross-K/V GEMM on the large-M form (gemm3_kernel) — A/B (WLX_CKV_GEMM3=0 = the 128 x 128 tile of the second form), small.en and large-v3.
set -u
TAG=${1:-r6k}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for m in small.en large-v3; do for v in 1 0 1 0; do
  echo "== $m WLX_CKV_GEMM3=$v"; WLX_CKV_GEMM3=$v timeout 300 python bench.py --model $m --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', round(d['value'],1), d['stage_ms'])"
done; done 2>&1 | tee "$OUT/ckv_gemm3_ab.txt"
timeout 600 python -m pytest tests/test_gpu_encoder_batched.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -3
