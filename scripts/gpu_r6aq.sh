#!/bin/bash
# Log G5 as adopted: wide K slices by default (12 / 10 / 8 k-tiles per wave where the waves spread evenly over the SIMDs). Every GPU test on libwlx.so,
# then WLX_G2_CHMAX=6 (the pick before) against the default on libwlx_ab.so, alternating: small.en, medium.en, large-v3, base.en, four streams, 12 windows per decode.
set -u
TAG=${1:-r6aq}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 --tb=short 2>&1 | tail -40 > "$OUT/pytest.log"; tail -2 "$OUT/pytest.log"
export WLX_LIB=whisperlive_amd/libwlx_ab.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3), 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"; }
for i in 1 2 3; do
  for C in 12 6; do
    WLX_G2_CHMAX=$C timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | line "small.en WLX_G2_CHMAX=$C"
  done
done | tee "$OUT/bench_ab.txt"
for M in large-v3 medium.en base.en; do
  for i in 1 2; do
    for C in 12 6; do
      WLX_G2_CHMAX=$C timeout 400 python bench.py --model $M --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 6 --warmup 2 2>/dev/null | line "$M WLX_G2_CHMAX=$C"
    done
  done
done | tee -a "$OUT/bench_ab.txt"
for i in 1 2; do
  for C in 12 6; do
    WLX_G2_CHMAX=$C timeout 400 python bench.py --streams 4 --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams 4 WLX_G2_CHMAX=$C', round(d['value'],1), 'mean', round(d['ms_per_step'],3))"
    WLX_G2_CHMAX=$C timeout 400 python bench.py --batch 12 --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 12 WLX_G2_CHMAX=$C', round(d['value'],1), 'mean', round(d['ms_per_step'],3))"
  done
done | tee -a "$OUT/bench_ab.txt"
