#!/bin/bash
# LDS-shared encoder attention (form 3): parity, A/B timing, rocprof stats of the encoder
set -u
TAG=${1:-r2n}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_depth.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "encoder or depth or layer" > "$OUT/pytest_enc.log" 2>&1; echo "pytest rc=$?"
tail -4 "$OUT/pytest_enc.log"
for cfg in "X=0" "WLX_ENC_ATTN=2" "X=1" "WLX_ENC_ATTN=2"; do
  echo -n "[$cfg] "; env $cfg timeout 120 python scripts/encode_only.py small.en 6 2>&1 | grep encode_ms
done
echo -n "[large-v3 new] "; timeout 300 python scripts/encode_only.py large-v3 3 2>&1 | grep encode_ms
echo -n "[large-v3 form 2] "; WLX_ENC_ATTN=2 timeout 300 python scripts/encode_only.py large-v3 3 2>&1 | grep encode_ms
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof_enc" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 4 > "$OUT/rocprof_enc.log" 2>&1; echo "rocprof enc rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof_enc" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && head -10 "$F" | cut -c1-180
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
