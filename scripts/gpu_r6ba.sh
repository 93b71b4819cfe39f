#!/bin/bash
# Log G11: the K-split MLP output projection of d_model 1024 / 1280 as four waves of two chunks instead of eight waves of one (WLX_G2_SLAB_NW4=1 on
# libwlx_ab.so), alternating: large-v3, medium.en.
set -u
TAG=${1:-r6ba}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=whisperlive_amd/libwlx_ab.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3), 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"; }
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
for M in large-v3 medium.en; do
  for i in 1 2; do
    for V in 1 0; do
      WLX_G2_SLAB_NW4=$V timeout 400 $B --model $M --steps 6 --warmup 2 2>/dev/null | line "$M WLX_G2_SLAB_NW4=$V"
    done
  done
done | tee "$OUT/bench_ab.txt"
