#!/bin/bash
# Log S9: two decode steps per graph launch where one slot is live on the device (engine.hip graph_steps_for), A/B on libwlx_ab.so with
# WLX_GRAPH_STEPS=1 / 2: every GPU test on the default library, headline (alternating), the stream leg at 16 and 64 tokens per chunk.
set -u
TAG=${1:-r6af}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=900 2>&1 | tail -15 | tee "$OUT/pytest_tail.txt"
for i in 1 2 3; do
  for N in 2 1; do
    WLX_LIB=whisperlive_amd/libwlx_ab.so WLX_GRAPH_STEPS=$N timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WLX_GRAPH_STEPS=$N', round(d['value'],1), round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), d['stage_ms'])"
  done
done | tee "$OUT/bench_ab.txt"
for N in 2 1; do
  for T in 16 64; do
    WLX_LIB=whisperlive_amd/libwlx_ab.so WLX_GRAPH_STEPS=$N timeout 600 python bench.py --no-cpu-baseline --no-throughput --no-pmc --steps 10 --warmup 3 --decode-steps $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('stream') or {}; print('WLX_GRAPH_STEPS=$N tokens=$T stream', {k: s.get(k) for k in ('p50_chunk_latency_ms','p95_chunk_latency_ms','xrt_per_stream','paced','stage_ms_per_chunk')})"
  done
done | tee "$OUT/stream_ab.txt"
