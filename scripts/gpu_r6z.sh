#!/bin/bash
# Round 6: the beam search pair after the rework of search_merge_update3_kernel (one trip of loads at entry, bookkeeping one candidate per lane):
# the exact search tests, the in-kernel timeline of a step (libwlx_trace.so), the headline.
set -u
TAG=${1:-r6z}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean_family.py tests/test_gpu_long_context.py tests/test_gpu_full_depth.py tests/test_gpu_batched_depth.py tests/test_gpu_transcriber.py -m gpu -q -p no:cacheprovider --timeout=900 2>&1 | tail -5 | tee "$OUT/pytest_tail.txt"
WLX_LIB=whisperlive_amd/libwlx_trace.so timeout 300 python scripts/trace_step.py --model small.en --t 33 > "$OUT/decode_step_trace.txt" 2>&1; tail -5 "$OUT/decode_step_trace.txt"
for i in 1 2; do timeout 300 python scripts/step_by_position.py small.en 2>&1 | tail -1; done | tee "$OUT/step_by_position.txt"
timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), d['stage_ms'])" | tee "$OUT/bench.txt"
