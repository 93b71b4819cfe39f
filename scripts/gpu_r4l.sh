#!/bin/bash
# round 4, call L: log-mel without the per-item host wait — parity of everything that touches features, batch-12 bench
set -u
TAG=${1:-r4l}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_jfk_fixture.py tests/test_gpu_transcriber.py tests/test_gpu_encoder_batched.py tests/test_trained_tiny.py -m gpu -q -x -p no:cacheprovider --timeout=800 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"
for i in 1 2; do timeout 600 python bench.py --batch 12 --steps 6 --warmup 2 --no-stream --no-cpu-baseline --no-pmc > "$OUT/bench_batch12_$i.json" 2> "$OUT/bench_batch12_$i.err"; python -c "
import json; d=json.loads(open('$OUT/bench_batch12_$i.json').read().strip().splitlines()[-1]); print('batch12', round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms'])"; done
timeout 600 python bench.py --steps 10 --warmup 3 --no-stream --no-cpu-baseline --no-pmc --no-throughput > "$OUT/bench.json" 2> "$OUT/bench.err"; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('single', round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms'])"
