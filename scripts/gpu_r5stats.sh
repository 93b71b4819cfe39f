#!/bin/bash
# rocprofv3 kernel stats of the driver's bench command on the final tree (the bench's own child legs switched off: no nested profilers)
set -u
TAG=${1:-r5s}; REPO=$PWD; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > "$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
tail -1 "$OUT/rocprof.log" | cut -c1-300
f=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_bench_command.csv" && head -12 "$f" | cut -c1-160
find "$OUT/rocprof" -name '*kernel_trace.csv' -delete; find "$OUT/rocprof" -name '*.db' -delete
du -sh "$OUT"
