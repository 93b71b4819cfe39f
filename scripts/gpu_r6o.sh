#!/bin/bash
# Round 6: per-kernel time of the one-window encoder after the full-line activation pieces (rocprofv3 --kernel-trace --stats, small.en and large-v3).
set -u
TAG=${1:-r6o}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
REPO=$PWD; cd /tmp
for m in small.en large-v3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$m" -o enc --output-format csv -- python $REPO/scripts/encode_only.py $m 10 1 > /dev/null 2>&1
  f=$(find "$OUT/prof_$m" -name '*kernel_stats.csv' | head -1); echo "== $m"; head -16 "$f" | cut -c1-220
done 2>&1 | tee "$OUT/encoder_kernel_stats.txt"
