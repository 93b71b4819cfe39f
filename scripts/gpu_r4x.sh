#!/bin/bash
# round 4, call X: two executables of the step graph replayed alternately (graph-to-graph boundary experiment)
set -u
TAG=${1:-r4x}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
for P in 0 1 0 1; do WLX_GRAPH_PAIR=$P timeout 600 python bench.py --steps 20 --warmup 3 --no-stream --no-cpu-baseline --no-pmc --no-throughput 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$P single', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()}, round(d['decode_step']['graph_replay_ms'],4))"; done | tee "$OUT/graph_pair.txt"
