#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel stats of the same command.
# usage: scripts/gpu_bench_profile.sh <tag> [extra bench args]
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -c 3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/rocprof.log" 2>&1
echo "rocprof rc=$?"
cd "$REPO"
find "$OUT/rocprof" -name '*kernel_stats*' | head
F=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && head -40 "$F"
# keep only the summaries (traces are large)
find "$OUT/rocprof" -name '*kernel_trace.csv' -size +20M -delete
du -sh "$OUT"
