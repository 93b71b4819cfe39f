#!/bin/bash
# round 4, call Y: rocprofv3 kernel stats of the 12-windows-per-decode bench (what the search pair costs at 60 rows)
set -u
TAG=${1:-r4y}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1; REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- python "$REPO/bench.py" --batch 12 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-stream > "$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
F=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp "$F" "$OUT/kernel_stats_batch12.csv" && head -22 "$F" | cut -c1-150
find "$OUT" -name '*kernel_trace.csv' -size +1M -delete
