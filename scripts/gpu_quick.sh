#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, real-kernel chain microbenchmark, short bench, decode-step trace.
# usage: scripts/gpu_quick.sh <tag>
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --timeout=240 > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -5 $OUT/pytest.log
timeout 100 scripts/ubench/chain2 > $OUT/chain2.txt 2>&1; cat $OUT/chain2.txt
timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('xRT', d['value'], 'ms/window', d['ms_per_step'], d['stage_ms'], 'step graph ms', d['decode_step']['graph_replay_ms'])" || tail -5 $OUT/bench.err
timeout 200 python scripts/trace_step.py --csv $OUT/trace.csv > $OUT/trace.txt 2>&1; head -10 $OUT/trace.txt; tail -4 $OUT/trace.txt
