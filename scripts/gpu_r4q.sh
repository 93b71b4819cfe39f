#!/bin/bash
# round 4, call Q: two column tiles per workgroup for the row-tiled wide LayerNorm projections (WLX_RT_NTB2) — parity, steps, benches, A/B
set -u
TAG=${1:-r4q}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_batched_depth.py -m gpu -q -x -p no:cacheprovider --timeout=1100 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
prof() { env $1 WLX_QUIET=1 timeout 600 python scripts/step_profile.py $2 $3 33 2>&1 | tee -a "$OUT/steps.txt" | head -${4:-3}; }
prof WLX_RT_NTB2=1 small.en 60 12
prof WLX_RT_NTB2=0 small.en 60 3
prof WLX_RT_NTB2=1 small.en 40 3
prof WLX_RT_NTB2=0 small.en 40 3
prof WLX_RT_NTB2=1 small.en 20 3
prof WLX_RT_NTB2=0 small.en 20 3
run() { env $1 WLX_QUIET=1 timeout 600 python bench.py $2 --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['stage_ms'].items()})"; }
{
run WLX_RT_NTB2=1 "--streams 4 --batch 12 --steps 6 --warmup 1 --free-run"
run WLX_RT_NTB2=0 "--streams 4 --batch 12 --steps 6 --warmup 1 --free-run"
run WLX_RT_NTB2=1 "--batch 12 --steps 4 --warmup 2"
run WLX_RT_NTB2=0 "--batch 12 --steps 4 --warmup 2"
run WLX_RT_NTB2=1 "--streams 4 --batch 12 --steps 6 --warmup 1 --free-run"
run WLX_RT_NTB2=0 "--streams 4 --batch 12 --steps 6 --warmup 1 --free-run"
} | tee "$OUT/bench_ab.txt"
