#!/bin/bash
# round 4, call H: gemm3 timing probes (no epilogue / K loop cut to one pair) and the per-launch sequence of one encoder layer
set -u
TAG=${1:-r4h}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
enc() { env "$1" timeout 600 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s/^/$1 /"; }
{
enc WLX_GEMM3_PROBE=0 small.en 12
enc WLX_GEMM3_PROBE=1 small.en 12
enc WLX_GEMM3_PROBE=2 small.en 12
enc WLX_GEMM3_PROBE=3 small.en 12
enc WLX_GEMM3_PROBE=0 large-v3 8
enc WLX_GEMM3_PROBE=1 large-v3 8
enc WLX_GEMM3_PROBE=2 large-v3 8
} | tee "$OUT/encode_probe_times.txt"
cd /tmp
for PR in 0 1 2; do
  D="$OUT/rp_small_12_probe$PR"
  WLX_GEMM3_PROBE=$PR timeout 600 rocprofv3 --kernel-trace -d "$D" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 12 > "$D.log" 2>&1; echo "rocprof rc=$?"
  python "$REPO/scripts/trace_sequence.py" "$D" "small.en B=12 probe=$PR" 44 8 | tee -a "$OUT/layer_sequence.txt"
done
find "$OUT" -name '*.csv' -size +1M -delete
