#!/bin/bash
# Log G9 again, with the second row of a wave requested together with its first (PF2 in dec_gemv2_kernel): the FIRST projection of a layer as FOUR waves
# (K = 768 four of six k-tiles, 1024 / 1280 four of 8 / 10) against one row per wave: WLX_G2_XS_FEW=1 on libwlx_ab.so, alternating.
set -u
TAG=${1:-r6az}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=whisperlive_amd/libwlx_ab.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3), 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"; }
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
for i in 1 2 3; do
  for V in 1 0; do
    WLX_G2_XS_FEW=$V timeout 300 $B --steps 20 --warmup 5 2>/dev/null | line "small.en WLX_G2_XS_FEW=$V"
  done
done | tee "$OUT/bench_ab.txt"
for M in large-v3 medium.en; do
  for i in 1 2; do
    for V in 1 0; do
      WLX_G2_XS_FEW=$V timeout 400 $B --model $M --steps 6 --warmup 2 2>/dev/null | line "$M WLX_G2_XS_FEW=$V"
    done
  done
done | tee -a "$OUT/bench_ab.txt"
WLX_G2_XS_FEW=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lean_family.py tests/test_gpu_full_depth.py tests/test_trained_tiny.py -m gpu -q -p no:cacheprovider --timeout=900 --tb=short 2>&1 | tail -8 | tee "$OUT/pytest_xs_few.txt"
