#!/bin/bash
# Log G4: wave-local combine of the cross-attention output projection with the (m, l) of the splits through LDS (one load per wave);
# A/B against libwlx_ml0.so (-DWLX_XCOMB_ML_LDS=0, same tree): decode parity tests, step graph by
# position (alternating), in-kernel timeline, headline (alternating).
set -u
TAG=${1:-r6ai}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean_family.py tests/test_gpu_long_context.py tests/test_gpu_full_depth.py tests/test_gpu_batched_depth.py tests/test_gpu_transcriber.py tests/test_gpu_finish.py tests/test_trained_tiny.py -m gpu -q -p no:cacheprovider --timeout=900 2>&1 | tail -25 | tee "$OUT/pytest_tail.txt"
for i in 1 2 3; do
  for L in libwlx.so libwlx_ml0.so; do WLX_LIB=whisperlive_amd/$L timeout 300 python scripts/step_by_position.py small.en 2>&1 | tail -1; done
done | tee "$OUT/step_by_position_ab.txt"
WLX_LIB=whisperlive_amd/libwlx_trace.so timeout 300 python scripts/trace_step.py --model small.en --t 33 > "$OUT/decode_step_trace.txt" 2>&1; sed -n 1,9p "$OUT/decode_step_trace.txt" | cut -c1-170
for i in 1 2 3 4; do
  for L in libwlx.so libwlx_ml0.so; do
    WLX_LIB=whisperlive_amd/$L timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'p50', round(d['p50_chunk_latency_ms'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3))"
  done
done | tee "$OUT/bench_ab.txt"
