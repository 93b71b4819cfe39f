#!/bin/bash
# Last check of the round on the binaries as they will be handed over: every GPU test, smoke(), the driver's bench command.
set -u
TAG=${1:-r6last}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -1 "$OUT/pytest.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
python -c "import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value','value_conditioned','ms_per_step','p50_chunk_latency_ms')}, d['roofline']['frac'], d['cpu_baseline']['value'])"
