#!/bin/bash
# Log G8: LayerNorm-fronted projections on PLAIN rows of one stream's step (cross-attention query, first MLP projection) as four waves of 8 / 10 k-tiles
# instead of eight of 4 / 5 for K = 1024 / 1280 (WLX_G2_LN_WIDE=1 on libwlx_ab.so), alternating: large-v3, medium.en; parity files with it on.
set -u
TAG=${1:-r6aw}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=whisperlive_amd/libwlx_ab.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3), 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"; }
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
for i in 1 2; do
  for V in 1 0; do
    WLX_G2_LN_WIDE=$V timeout 400 $B --model large-v3 --steps 6 --warmup 2 2>/dev/null | line "large-v3 WLX_G2_LN_WIDE=$V"
  done
done | tee "$OUT/bench_ab.txt"
for i in 1 2; do
  for V in 1 0; do
    WLX_G2_LN_WIDE=$V timeout 400 $B --model medium.en --steps 6 --warmup 2 2>/dev/null | line "medium.en WLX_G2_LN_WIDE=$V"
  done
done | tee -a "$OUT/bench_ab.txt"
WLX_G2_LN_WIDE=1 timeout 900 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_full_depth.py tests/test_gpu_batched_depth.py -m gpu -q -p no:cacheprovider --timeout=900 --tb=short 2>&1 | tail -8 | tee "$OUT/pytest_ln_wide.txt"
