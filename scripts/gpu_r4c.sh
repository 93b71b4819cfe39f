#!/bin/bash
# round 4, call C: the dedicated vocabulary projection (dec_vocab_kernel) — parity, then timing at 5 / 20 / 40 / 60 rows; row-tile
# policy for the wide projections with the K split off
set -u
TAG=r4c; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_batched_depth.py tests/test_gpu_full_depth.py -m gpu -q -x -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
run() { # env... -- model rows
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python scripts/step_profile.py "$@" 2>&1 | tee -a "$OUT/steps.txt" | head -${LINES_OUT:-3}
}
LINES_OUT=14
run A=1 -- small.en 5
run WLX_VOCAB2=0 -- small.en 5
run A=1 -- small.en 60
LINES_OUT=4
run A=1 -- small.en 40
run A=1 -- small.en 20
run WLX_ROWTILE_NMAX=1000 -- small.en 20
run WLX_ROWTILE=0 -- small.en 20
LINES_OUT=14
run A=1 -- large-v3 40
LINES_OUT=6
run WLX_ROWTILE_NMAX=1300 -- large-v3 40
run WLX_ROWTILE_NMAX=3900 -- large-v3 40
run A=1 -- large-v3 5
run WLX_VOCAB2=0 -- large-v3 5
timeout 600 python bench.py --batch 12 --steps 4 --warmup 2 --no-stream --no-cpu-baseline --no-pmc > "$OUT/bench_batch12.json" 2> "$OUT/bench_batch12.err"; python -c "
import json,sys; d=json.loads(open('$OUT/bench_batch12.json').read().strip().splitlines()[-1]); print('batch12', d['value'], d['ms_per_step'], d['stage_ms'], d['decode_step']['graph_replay_ms'])"
timeout 600 python bench.py --steps 10 --warmup 3 --no-stream --no-cpu-baseline --no-pmc > "$OUT/bench.json" 2> "$OUT/bench.err"; python -c "
import json,sys; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('single', d['value'], d['ms_per_step'], d['stage_ms'], d['decode_step']['graph_replay_ms'], d.get('parity_prefix'))"
