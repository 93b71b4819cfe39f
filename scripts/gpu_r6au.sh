#!/bin/bash
# Log G5, large-v3 (configs[3] runs it one stream per GPU): K = 1280 as four waves of ten k-tiles (WLX_G2_CH10=1 on libwlx_ab.so) against eight of five.
set -u
TAG=${1:-r6au}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=whisperlive_amd/libwlx_ab.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'mean', round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), 'generate', round(d['stage_ms']['generate_ms'],3), 'step', round(1e3*d['decode_step']['graph_replay_ms'],1))"; }
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
for i in 1 2 3; do
  for V in 1 0; do
    WLX_G2_CH10=$V timeout 400 $B --model large-v3 --steps 6 --warmup 2 2>/dev/null | line "large-v3 WLX_G2_CH10=$V"
  done
done | tee "$OUT/bench_ab.txt"
WLX_G2_CH10=1 timeout 900 python -m pytest tests/test_gpu_batched_depth.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=900 --tb=short 2>&1 | tail -30 | tee "$OUT/pytest_ch10.txt"
