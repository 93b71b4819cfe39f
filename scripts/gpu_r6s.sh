#!/bin/bash
# Round 6: the self-attention with whole-row K fetches on the headline — bench A/B against libwlx_klane.so (previous kernel), then every GPU test.
set -u
TAG=${1:-r6s}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for lib in libwlx.so libwlx_klane.so libwlx.so libwlx_klane.so; do
  echo -n "$lib  "; WLX_LIB=whisperlive_amd/$lib timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), 'conditioned', round(d.get('value_conditioned') or 0,1), d['stage_ms'])"
done 2>&1 | tee "$OUT/bench_ab.txt"
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > "$OUT/pytest_full.log" 2>&1; tail -3 "$OUT/pytest_full.log"
