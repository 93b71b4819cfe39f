#!/bin/bash
# Round 6: decode self-attention with 8 waves per (row, head) (one block of 64 positions per wave) against 4 (libwlx_sa4.so = -DSA_NW=4): step by
# position, prompt prefill, the bench line's conditioned window; then every GPU test.
set -u
TAG=${1:-r6w}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for lib in libwlx.so libwlx_sa4.so libwlx.so libwlx_sa4.so; do WLX_LIB=whisperlive_amd/$lib timeout 300 python scripts/step_by_position.py small.en 2>&1 | tail -1; done | tee "$OUT/step_by_position_sa8.txt"
for lib in libwlx.so libwlx_sa4.so libwlx.so libwlx_sa4.so; do echo "== $lib"; WLX_LIB=whisperlive_amd/$lib timeout 300 python scripts/prefill_time.py 2>&1 | grep -i "prefill" | tail -3; done | tee "$OUT/prefill_sa8.txt"
for lib in libwlx.so libwlx_sa4.so libwlx.so libwlx_sa4.so; do
  echo -n "$lib  "; WLX_LIB=whisperlive_amd/$lib timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1))"
done 2>&1 | tee "$OUT/bench_sa8.txt"
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > "$OUT/pytest_full.log" 2>&1; tail -3 "$OUT/pytest_full.log"
