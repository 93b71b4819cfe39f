#!/bin/bash
# The handed-over binaries once more after the last (comment-only) rebuild: smoke(), the parity files, the driver's bench command without its CPU leg.
set -u
TAG=${1:-r6last2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_depth.py tests/test_gpu_lean_family.py tests/test_gpu_batched_depth.py -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > "$OUT/bench.json" 2>/dev/null; python -c "import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value','value_conditioned','ms_per_step','p50_chunk_latency_ms')})"
