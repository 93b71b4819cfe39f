#!/bin/bash
# Log G5 continued, on a libwlx_ab.so that is really this tree's (r6aq / r6ar measured the previous session's: build_all.sh was given a bare variant name):
# K = 768 as ONE wave of 24 k-tiles against two of twelve; K = 1024 as two waves of sixteen against four of eight; the split combine (cross-attention
# output projection) as three waves of eight / two of twelve (three / four items peeled per lane) against four of six; the rule's no-change cases.
set -u
TAG=${1:-r6as}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=whisperlive_amd/libwlx_ab.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3), 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"; }
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
for i in 1 2 3; do
  for V in "WLX_G2_CHMAX=24" "WLX_G2_CHMAX=12" "WLX_G2_XCHMAX=8" "WLX_G2_XCHMAX=12" "WLX_G2_CHMAX=6"; do
    env $V timeout 300 $B --steps 20 --warmup 5 2>/dev/null | line "small.en $V"
  done
done | tee "$OUT/bench_ab.txt"
for i in 1 2; do
  for V in "WLX_G2_CHMAX=24" "WLX_G2_CHMAX=12" "WLX_G2_CHMAX=6"; do
    env $V timeout 400 $B --model medium.en --steps 6 --warmup 2 2>/dev/null | line "medium.en $V"
  done
done | tee -a "$OUT/bench_ab.txt"
for V in "WLX_G2_CHMAX=12" "WLX_G2_CHMAX=6"; do
  env $V timeout 400 $B --model large-v3 --steps 6 --warmup 2 2>/dev/null | line "large-v3 $V"
  env $V timeout 400 $B --batch 12 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 12 $V', round(d['value'],1), 'mean', round(d['ms_per_step'],3))"
done | tee -a "$OUT/bench_ab.txt"
WLX_G2_CHMAX=24 WLX_G2_XCHMAX=8 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_depth.py tests/test_gpu_lean_family.py tests/test_trained_tiny.py -m gpu -q -p no:cacheprovider --timeout=600 --tb=short 2>&1 | tail -15 | tee "$OUT/pytest_chmax24_xchmax8.txt"
