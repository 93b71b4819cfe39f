#!/bin/bash
# Log G1: dec_cq_cross_attn_kernel with LayerNorm gamma / beta through LDS (one KiB piece per wave instead of all six per wave) and the softmax
# reductions over the four lane rows by v_permlane swaps; A/B against libwlx_v1.so (-DWLX_CQ_SWAP=0) and libwlx_v0.so (both off), same tree:
# parity tests, step graph by position (alternating), in-kernel timeline, headline bench (alternating).
# (First run of this script: the gamma / beta exchange in dec_gemv2_kernel's LayerNorm prologues as well — scripts/patches/r6ad_*.diff — +11 us per step.)
set -u
TAG=${1:-r6ad}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean_family.py tests/test_gpu_long_context.py tests/test_gpu_full_depth.py tests/test_gpu_batched_depth.py tests/test_gpu_transcriber.py -m gpu -q -x -p no:cacheprovider --timeout=900 2>&1 | tail -25 | tee "$OUT/pytest_tail.txt"
for i in 1 2 3; do
  for L in libwlx.so libwlx_v1.so libwlx_v0.so; do WLX_LIB=whisperlive_amd/$L timeout 300 python scripts/step_by_position.py small.en 2>&1 | tail -1; done
done | tee "$OUT/step_by_position_ab.txt"
WLX_LIB=whisperlive_amd/libwlx_trace.so timeout 300 python scripts/trace_step.py --model small.en --t 33 > "$OUT/decode_step_trace.txt" 2>&1; tail -3 "$OUT/decode_step_trace.txt"
for i in 1 2; do
  for L in libwlx.so libwlx_v1.so libwlx_v0.so; do
    WLX_LIB=whisperlive_amd/$L timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],1), round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), d['stage_ms'])"
  done
done | tee "$OUT/bench_ab.txt"
