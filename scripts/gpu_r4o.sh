#!/bin/bash
# round 4, call O: 4 x 12 windows — launch-count / workgroup-count knobs under concurrency
set -u
TAG=${1:-r4o}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
run() { env $1 timeout 600 python bench.py --streams 4 --batch 12 --steps 6 --warmup 1 --free-run --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms'].items()})"; }
{
run A=1
run WLX_XATTN_SEPARATE=0
run WLX_FC2_KS_BATCHED=1
run WLX_SLOT_CU_MASK=off
run A=2
} | tee "$OUT/s4_b12_knobs.txt"
