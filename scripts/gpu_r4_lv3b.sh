#!/bin/bash
# Round 4: large-v3 batched step with the K-split MLP output projection kept (WLX_FC2_KS_BATCHED=1), and config 5 at 12 clips per decode
set -u
TAG=${1:-r4lv3b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
prof() { env $1 timeout 600 python scripts/step_profile.py $2 $3 33 2>&1 | sed "s/^==/== [$1]/" | tee -a "$OUT/steps.txt" | head -${4:-1}; }
prof A=1 large-v3 40 10
prof WLX_FC2_KS_BATCHED=1 large-v3 40 10
run() { env $1 timeout 900 python bench.py $2 --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$1] $2', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a "$OUT/bench_ab.txt"; }
run A=1 "--config 5 --lanes 1 --max-batch 12 --steps 2 --warmup 1"
run A=1 "--config 5 --max-batch 12 --steps 2 --warmup 1"
echo done
