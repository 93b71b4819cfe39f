#!/bin/bash
# Round 6: the one-pass prompt prefill without the K-split MLP projection (its slabs make every 16-column workgroup of the next layer's first projection
# read three fp32 copies of its rows) — A/B against WLX_PREFILL_SLABS=1, then the long-context parity tests on the new default.
set -u
TAG=${1:-r6f}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for v in 0 1 0 1; do echo "== WLX_PREFILL_SLABS=$v"; WLX_PREFILL_SLABS=$v timeout 200 python scripts/prefill_time.py small.en 2>&1 | tail -3; done | tee "$OUT/prefill_time_small.txt"
for v in 0 1; do echo "== WLX_PREFILL_SLABS=$v large-v3"; WLX_PREFILL_SLABS=$v timeout 300 python scripts/prefill_time.py large-v3 2>&1 | tail -3; done | tee "$OUT/prefill_time_large_v3.txt"
timeout 900 python -m pytest tests/test_gpu_long_context.py tests/test_gpu_transcriber.py tests/test_trained_tiny.py -m gpu -q -p no:cacheprovider --timeout=600 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)|common prefix" "$OUT/pytest.log" | head -30
