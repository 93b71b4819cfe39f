#!/bin/bash
# Log G5: waves per workgroup of the fp16-rows-in projections now that no barrier stands in front of their MFMAs (libwlx_ab.so, WLX_G2_CHMAX): K = 768 as
# four waves of six k-tiles (the pick so far), three of eight, two of twelve (the K-split MLP projection: eight / six / four waves); first run: six waves of four.
set -u
TAG=${1:-r6al}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=whisperlive_amd/libwlx_ab.so
for i in 1 2 3; do
  for C in 24 12; do echo -n "WLX_G2_CHMAX=$C "; WLX_G2_CHMAX=$C timeout 300 python scripts/step_by_position.py small.en 2>&1 | tail -1; done
done | tee "$OUT/step_by_position_ab.txt"
for i in 1 2 3; do
  for C in 24 12; do
    WLX_G2_CHMAX=$C timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WLX_G2_CHMAX=$C', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'p50', round(d['p50_chunk_latency_ms'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3))"
  done
done | tee "$OUT/bench_ab.txt"
WLX_G2_CHMAX=24 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_depth.py -m gpu -q -x -p no:cacheprovider --timeout=900 2>&1 | tail -4 | tee "$OUT/pytest_tail_chmax24.txt"
