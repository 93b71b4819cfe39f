#!/bin/bash
# round 4, call I: gemm3 with register-direct V epilogues (transposed accumulators) and staged row-major epilogues — parity + times
set -u
TAG=${1:-r4i}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_encoder_batched.py -m gpu -q -x -s -p no:cacheprovider --timeout=800 > "$OUT/pytest_batched.log" 2>&1; echo "pytest batched rc=$?"; grep -E "passed|failed|max rel|Error|assert" "$OUT/pytest_batched.log" | head -12
WLX_GEMM3=2 timeout 900 python -m pytest tests/test_gpu_full_depth.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=800 -k "encoder or generate_beam or align" > "$OUT/pytest_forced.log" 2>&1; echo "pytest forced rc=$?"; tail -3 "$OUT/pytest_forced.log"
timeout 1200 python -m pytest tests/test_gpu_batched_depth.py -m gpu -q -x -p no:cacheprovider --timeout=1100 > "$OUT/pytest_lv3.log" 2>&1; echo "pytest lv3 rc=$?"; tail -3 "$OUT/pytest_lv3.log"
enc() { env "$1" timeout 600 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s/^/$1 /"; }
{
for B in 12 8 4; do enc WLX_GEMM3=1 small.en $B; done
enc WLX_GEMM3_PROBE=1 small.en 12
enc WLX_GEMM3=1 large-v3 8
enc WLX_GEMM3=1 large-v3 4
} | tee "$OUT/encode_times.txt"
cd /tmp
D="$OUT/rp_small_12"
timeout 600 rocprofv3 --kernel-trace -d "$D" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 12 > "$D.log" 2>&1; echo "rocprof rc=$?"
python "$REPO/scripts/trace_sequence.py" "$D" "small.en B=12" 44 8 | tee -a "$OUT/layer_sequence.txt"
python "$REPO/scripts/trace_by_grid.py" "$D" "small.en B=12" | tee -a "$OUT/layer_sequence.txt"
find "$OUT" -name '*.csv' -size +1M -delete
