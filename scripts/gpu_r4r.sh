#!/bin/bash
# round 4, call R: wider workgroups for row-tiled projections (4 column tiles for the LayerNorm-fronted wide ones, 2 for the fp16-rows-in residual ones) — A/B
set -u
TAG=${1:-r4r}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
prof() { env $1 timeout 600 python scripts/step_profile.py $2 $3 33 2>&1 | tee -a "$OUT/steps.txt" | head -${4:-3}; }
prof A=1 small.en 60 3
prof WLX_RT_NTB4=1 small.en 60 8
prof WLX_RT_F16_NTB2=1 small.en 60 8
prof "WLX_RT_NTB4=1 WLX_RT_F16_NTB2=1" small.en 40 3
prof A=1 small.en 40 3
run() { env $1 timeout 600 python bench.py $2 --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms'].items()})"; }
{
run A=1 "--streams 4 --batch 12 --steps 6 --warmup 1 --free-run"
run WLX_RT_NTB4=1 "--streams 4 --batch 12 --steps 6 --warmup 1 --free-run"
run WLX_RT_F16_NTB2=1 "--streams 4 --batch 12 --steps 6 --warmup 1 --free-run"
run "WLX_RT_NTB4=1 WLX_RT_F16_NTB2=1" "--streams 4 --batch 12 --steps 6 --warmup 1 --free-run"
run A=2 "--streams 4 --batch 12 --steps 6 --warmup 1 --free-run"
} | tee "$OUT/bench_ab.txt"
timeout 900 env WLX_RT_NTB4=1 WLX_RT_F16_NTB2=1 python -m pytest tests/test_gpu_lean_family.py -m gpu -q -x -p no:cacheprovider --timeout=800 -k "twelve or eight or rows" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
