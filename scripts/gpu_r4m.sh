#!/bin/bash
# round 4, call M: the row-tile policy under 4-slot concurrency (4 x 12 windows): more, smaller workgroups vs fewer, larger ones
set -u
TAG=${1:-r4m}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
run() { env $1 timeout 600 python bench.py --streams 4 --batch 12 --steps 3 --warmup 1 --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms'].items()})"; }
{
run A=1
run WLX_ROWTILE=0
run WLX_ROWTILE_CHUNK=32
run WLX_ROWTILE_NMAX=1000
run WLX_VOCAB2=0
run A=2
} | tee "$OUT/s4_b12_policy.txt"
