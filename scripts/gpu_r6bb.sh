#!/bin/bash
# rocprofv3 --kernel-trace --stats of the large-v3 single-stream bench (configs[3]'s per-GPU workload) on the final tree: per-kernel averages after logs G5 / G8 / G9.
set -u
TAG=${1:-r6bb}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- python "$REPO/bench.py" --model large-v3 --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 4 --warmup 1 > "$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
f=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_large_v3_bench.csv" && head -16 "$f" | cut -c1-170
find "$OUT/rocprof" -name '*kernel_trace.csv' -delete; find "$OUT/rocprof" -name '*.db' -delete
tail -1 "$OUT/rocprof.log" | cut -c1-300
