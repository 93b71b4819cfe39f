#!/bin/bash
# round 4, call N: 4 x 12 windows with phase-shifted (free-running) streams vs lockstep; determinism stress test of the batched encoder
set -u
TAG=${1:-r4n}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder_batched.py -m gpu -q -x -p no:cacheprovider --timeout=800 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
run() { timeout 600 python bench.py "$@" --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value'],1), round(d['ms_per_step'],2), 'p50', round(d['p50_chunk_latency_ms'],1), {k: round(v,2) for k,v in d['stage_ms'].items()})"; }
{
run --streams 4 --batch 12 --steps 4 --warmup 1
run --streams 4 --batch 12 --steps 4 --warmup 1 --free-run
run --streams 4 --batch 12 --steps 8 --warmup 1 --free-run
run --streams 3 --batch 12 --steps 6 --warmup 1 --free-run
run --streams 2 --batch 12 --steps 6 --warmup 1 --free-run
run --streams 4 --steps 12 --warmup 2
run --streams 4 --steps 12 --warmup 2 --free-run
} | tee "$OUT/free_run.txt"
