#!/bin/bash
# Round 6, first GPU call: every GPU test (tightened tolerances, reference log-mel vectors, RCCL at world size 1) + the default bench line.
set -u
TAG=${1:-r6a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 "$OUT/pytest.log"
grep -E "FAILED|ERROR" "$OUT/pytest.log" | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log" | cut -c1-300
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"; tail -3 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "value_conditioned", "ms_per_step", "p50_chunk_latency_ms", "stage_ms", "h2d_excluded_ms")})
print("roofline:", json.dumps({k: v for k, v in d["roofline"].items() if k not in ("largest_launch", "traffic_detail")}))
print("cond:", d.get("conditioned_window"))
print("cpu:", json.dumps(d.get("cpu_baseline"))[:1500]); print("parity:", d.get("parity_prefix")); print("stream:", json.dumps(d.get("stream", {}).get("unpaced")), json.dumps(d.get("stream", {}).get("paced_256ms")))
print("throughput:", json.dumps(d.get("throughput")))
PY
echo "total $(( $(date +%s) - t0 )) s"
