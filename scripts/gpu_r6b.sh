#!/bin/bash
# Round 6, second GPU call: the PCM ring and the 8-wave long-context self-attention — their tests, the default bench line (conditioned window, stream leg
# through the ring), the decode step by position.
set -u
TAG=${1:-r6b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_long_context.py tests/test_gpu_transcriber.py tests/test_server.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)" "$OUT/pytest.log" | head -20
timeout 900 python bench.py --no-cpu-baseline --no-throughput > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"; tail -3 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "value_conditioned", "ms_per_step", "p50_chunk_latency_ms", "stage_ms", "h2d_excluded_ms")})
print("cond:", d.get("conditioned_window"))
print("stream:", json.dumps(d.get("stream", {}).get("unpaced")), json.dumps(d.get("stream", {}).get("paced_256ms")))
PY
timeout 300 python scripts/step_by_position.py > "$OUT/step_by_position.txt" 2>&1; cat "$OUT/step_by_position.txt" | tail -12
WLX_PCM_RING=0 timeout 600 python bench.py --no-cpu-baseline --no-throughput --no-pmc --steps 5 > "$OUT/bench_noring.json" 2> "$OUT/bench_noring.err"; echo "bench (ring off) rc=$?"
python - "$OUT/bench_noring.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ring off stream:", json.dumps(d.get("stream", {}).get("unpaced")), json.dumps(d.get("stream", {}).get("paced_256ms")))
PY
echo "total $(( $(date +%s) - t0 )) s"
