#!/bin/bash
# After logs G5 / G8: the two sizes they moved, on libwlx.so of the final tree (single stream, 30 s window, beam 5, 64 tokens).
set -u
TAG=${1:-r6final_b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
for M in large-v3 medium.en base.en; do
  timeout 400 $B --model $M --steps 6 --warmup 2 > "$OUT/bench_${M//./_}.json" 2>/dev/null
  python -c "import json,sys; d=json.loads(open('$OUT/bench_${M//./_}.json').read().strip().splitlines()[-1]); print('$M', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'conditioned', round(d.get('value_conditioned') or 0,1), d['stage_ms'], 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"
done | tee "$OUT/summary.txt"
timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --max-batch 16 --no-pmc > "$OUT/bench_config5_mb16.json" 2>/dev/null; python -c "import json; d=json.loads(open('$OUT/bench_config5_mb16.json').read().strip().splitlines()[-1]); print('config 5 mb16', round(d['value'],1), round(d['ms_per_step'],1))" | tee -a "$OUT/summary.txt"
