#!/bin/bash
# Log S9, second pass: is the one slow N = 2 run of r6af (1061.6 xRT mean, the others 1121-1122) the two-step graph or the box? Ten alternating pairs,
# mean AND p50 per run (bench.py: 20 windows each).
set -u
TAG=${1:-r6ag}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for i in 1 2 3 4 5 6 7 8 9 10; do
  for N in 2 1; do
    WLX_LIB=whisperlive_amd/libwlx_ab.so WLX_GRAPH_STEPS=$N timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WLX_GRAPH_STEPS=$N', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'p50', round(d['p50_chunk_latency_ms'],3), 'generate', round(d['stage_ms']['generate_ms'],3))"
  done
done | tee "$OUT/bench_ab.txt"
