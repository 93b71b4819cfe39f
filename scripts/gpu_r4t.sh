#!/bin/bash
# round 4, call T: more work-saving launch shapes under a busy device — A/B in the 4 x 12 configuration and config 5
set -u
TAG=${1:-r4t}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
run() { env $1 timeout 900 python bench.py $2 --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d.get('stage_ms',{}).items()})"; }
{
run A=1 "--streams 4 --batch 12 --steps 6 --warmup 1"
run WLX_RT_CQ_NTB2=1 "--streams 4 --batch 12 --steps 6 --warmup 1"
run WLX_RT_F16_NTB4=1 "--streams 4 --batch 12 --steps 6 --warmup 1"
run "WLX_RT_CQ_NTB2=1 WLX_RT_F16_NTB4=1" "--streams 4 --batch 12 --steps 6 --warmup 1"
run A=2 "--streams 4 --batch 12 --steps 6 --warmup 1"
run A=1 "--config 5 --steps 2 --warmup 1"
run WLX_RT_CQ_NTB2=1 "--config 5 --steps 2 --warmup 1"
run WLX_RT_F16_NTB4=1 "--config 5 --steps 2 --warmup 1"
} | tee "$OUT/bench_ab.txt"
