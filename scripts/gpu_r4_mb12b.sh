#!/bin/bash
set -u
TAG=${1:-r4mb12b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
timeout 100 python bench.py --config 5 --max-batch 12 --steps 1 --warmup 1 --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5, default lanes, 12 per decode:', round(d['value'],1), round(d['ms_per_step'],2))" | tee "$OUT/bench.txt"
