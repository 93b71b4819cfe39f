#!/bin/bash
# Log G6: K slices of the MLP output projection (WLX_FC2_KS, compile time) re-measured on top of the wide slices of log G5: 2 (the pick: 96 workgroups of
# four waves x 12 k-tiles for K = 3072) against 3 (144 x four waves of eight) and 4 (192 x two waves of twelve; the consumers add four slabs instead of two).
set -u
TAG=${1:-r6at}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3), 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"; }
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
for i in 1 2 3; do
  for L in libwlx.so libwlx_ks3.so libwlx_ks4.so; do
    WLX_LIB=whisperlive_amd/$L timeout 300 $B --steps 20 --warmup 5 2>/dev/null | line "small.en $L"
  done
done | tee "$OUT/bench_ab.txt"
for M in medium.en large-v3; do
  for L in libwlx.so libwlx_ks3.so libwlx_ks4.so; do
    WLX_LIB=whisperlive_amd/$L timeout 400 $B --model $M --steps 6 --warmup 2 2>/dev/null | line "$M $L"
  done
done | tee -a "$OUT/bench_ab.txt"
