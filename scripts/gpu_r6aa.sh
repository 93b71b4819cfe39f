#!/bin/bash
# Round 6: results of a generate in pinned host memory, read when the done word is seen (no copies, no wait for the step behind the finish):
# wall time of a decode that ends on an end-of-text, A/B against libwlx_prev.so (the parent commit's engine.hip / search.hip); headline; every GPU test.
set -u
TAG=${1:-r6aa}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for lib in libwlx.so libwlx_prev.so libwlx.so libwlx_prev.so; do echo -n "$lib  "; WLX_LIB=whisperlive_amd/$lib timeout 300 python scripts/finish_latency.py small.en 2>&1 | tail -1; done | tee "$OUT/finish_latency_ab.txt"
for lib in libwlx.so libwlx_prev.so; do
  echo -n "$lib  "; WLX_LIB=whisperlive_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['stream']; print(round(d['value'],1), round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'stream p50', round(st['unpaced']['p50_chunk_latency_ms'],2), round(st['paced_256ms']['p50_chunk_latency_ms'],2), st.get('stage_ms_per_chunk',{}).get('host_and_sync_ms'))"
done 2>&1 | tee "$OUT/bench_ab.txt"
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > "$OUT/pytest_full.log" 2>&1; tail -3 "$OUT/pytest_full.log"
