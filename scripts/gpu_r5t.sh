#!/bin/bash
# encoder LayerNorm A/B (DPP reductions, gamma/beta requested up front): rocprofv3 kernel stats of the encoder-only workload
set -u
TAG=${1:-r5t}; REPO=$PWD; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
cd /tmp
for m in small.en large-v3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/enc_$m" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" $m 20 1 > "$OUT/enc_$m.log" 2>&1; echo "$m rc=$?"; tail -1 "$OUT/enc_$m.log"
  f=$(find "$OUT/enc_$m" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/encoder_kernel_stats_$m.csv" && grep -i "layernorm\|Name" "$f" | cut -c1-200
  rm -rf "$OUT/enc_$m"
done
