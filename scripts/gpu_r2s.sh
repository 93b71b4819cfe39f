#!/bin/bash
# K-split MLP output projection for batched rows (M <= 48), no-spill multi-tile kernels: parity, config 5, 40-row trace
set -u
TAG=${1:-r2s}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_full_depth.py tests/test_gpu_transcriber.py -m gpu -q -x -p no:cacheprovider --timeout=600 > "$OUT/pytest_sub.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_sub.log"
timeout 300 python scripts/trace_step.py --model large-v3 --items 8 --t 20 --csv "$OUT/trace_l40.csv" > "$OUT/trace_l40.txt" 2>&1; echo "trace rc=$?"
sed -n 1,12p "$OUT/trace_l40.txt"; tail -3 "$OUT/trace_l40.txt"
for cfg in "X=0" "WLX_XATTN_SEPARATE=0"; do
env $cfg timeout 600 python bench.py --config 5 --steps 2 --warmup 1 > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"; echo "[$cfg] config5 rc=$?"
python -c "import json; d=json.loads(open('$OUT/bench_config5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['ms_per_step'], d.get('decode_step'))"
done
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stream --no-pmc > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"
python -c "import json; d=json.loads(open('$OUT/bench_quick.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['decode_step']['graph_replay_ms'])"
