#!/bin/bash
# Log G5, the other model sizes: WLX_G2_CHMAX=12 against 6 on libwlx_ab.so for tiny.en, base.en, medium.en, large-v3 (headline bench, alternating), and
# every GPU test with WLX_G2_CHMAX=12.
set -u
TAG=${1:-r6ao}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=whisperlive_amd/libwlx_ab.so
for M in tiny.en base.en medium.en large-v3; do
  for i in 1 2; do
    for C in 12 6; do
      WLX_G2_CHMAX=$C timeout 400 python bench.py --model $M --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$M WLX_G2_CHMAX=$C', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3), 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"
    done
  done
done | tee "$OUT/bench_ab.txt"
WLX_G2_CHMAX=12 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 2>&1 | tail -6 | tee "$OUT/pytest_tail_chmax12.txt"
