#!/bin/bash
# round 4, call W: wlx_encode without a host wait — parity subset, headline A/B is by run (two benches)
set -u
TAG=${1:-r4w}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_transcriber.py tests/test_gpu_full_depth.py tests/test_trained_tiny.py tests/test_gpu_lean_family.py -m gpu -q -x -p no:cacheprovider --timeout=1100 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 3 --no-stream --no-cpu-baseline --no-pmc --no-throughput 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()}, d['decode_step']['graph_replay_ms'])"; done
timeout 600 python bench.py --batch 12 --steps 4 --warmup 2 --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch12', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms'].items()})"
