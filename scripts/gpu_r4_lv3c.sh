#!/bin/bash
# Round 4: large-v3, 12 vs 8 windows per decode at the engine level (no worker): stage times
set -u
TAG=${1:-r4lv3c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
run() { timeout 900 python bench.py $1 --no-stream --no-cpu-baseline --no-pmc --no-throughput 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d.get('stage_ms',{}).items()}, 'step', round(d.get('decode_step',{}).get('graph_replay_ms',0),4))" | tee -a "$OUT/bench_ab.txt"; }
run "--model large-v3 --batch 12 --steps 3 --warmup 1"
run "--model large-v3 --batch 8 --steps 3 --warmup 1"
echo done
