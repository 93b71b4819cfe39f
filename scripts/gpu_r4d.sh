#!/bin/bash
# round 4, call D: the large-M encoder GEMM (gemm3_kernel) — parity (batched vs single-window encodes vs oracle; every GEMM of a
# single window forced through it with WLX_GEMM3=2), then the batched encoder's time and per-kernel table, on and off
set -u
TAG=r4d; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_encoder_batched.py -m gpu -q -x -s -p no:cacheprovider --timeout=800 > "$OUT/pytest_batched.log" 2>&1; echo "pytest batched rc=$?"; grep -E "passed|failed|max rel|Error|assert" "$OUT/pytest_batched.log" | head -12
WLX_GEMM3=2 timeout 900 python -m pytest tests/test_gpu_full_depth.py -m gpu -q -x -p no:cacheprovider --timeout=800 -k "encoder" > "$OUT/pytest_forced.log" 2>&1; echo "pytest forced rc=$?"; tail -3 "$OUT/pytest_forced.log"
for G in 1 0; do
  for B in 12 8 4; do WLX_GEMM3=$G timeout 300 python scripts/encode_only.py small.en 3 $B 2>&1 | tail -1 | sed "s/^/GEMM3=$G /"; done
  WLX_GEMM3=$G timeout 600 python scripts/encode_only.py large-v3 3 8 2>&1 | tail -1 | sed "s/^/GEMM3=$G /"
done | tee "$OUT/encode_times.txt"
WLX_GEMM3=2 timeout 300 python scripts/encode_only.py small.en 3 1 2>&1 | tail -1 | sed "s/^/GEMM3=2 /" | tee -a "$OUT/encode_times.txt"
cd /tmp
for G in 1 0; do
  WLX_GEMM3=$G timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof_g$G" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 3 12 > "$OUT/rocprof_g$G.log" 2>&1; echo "rocprof rc=$?"
  F=$(find "$OUT/rocprof_g$G" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp "$F" "$OUT/kernel_stats_b12_gemm3_$G.csv" && head -12 "$F" | cut -c1-170
done
find "$OUT" -name '*kernel_trace.csv' -size +1M -delete; find "$OUT" -name '*.csv' -path '*rocprof_g*' -size +2M -delete
