#!/bin/bash
# Log G10: the first MLP projection of Whisper-small (LayerNorm-fronted, PLAIN rows, K = 768) as three waves of eight k-tiles (two row trips, as now) /
# two waves of twelve (three row trips) against four of six: WLX_G2_LN768 on libwlx_ab.so, alternating.
set -u
TAG=${1:-r6ay}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=whisperlive_amd/libwlx_ab.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3), 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"; }
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
for i in 1 2 3; do
  for V in 8 12 0; do
    WLX_G2_LN768=$V timeout 300 $B --steps 20 --warmup 5 2>/dev/null | line "small.en WLX_G2_LN768=$V"
  done
done | tee "$OUT/bench_ab.txt"
WLX_G2_LN768=12 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 --tb=short 2>&1 | tail -4 | tee "$OUT/pytest_ln768_12.txt"
WLX_G2_LN768=8 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 --tb=short 2>&1 | tail -4 | tee "$OUT/pytest_ln768_8.txt"
