#!/bin/bash
# Final tree of round 5: every GPU test, smoke(), the default bench line exactly as the driver runs it.
set -u
TAG=${1:-r5end}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=300 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 "$OUT/pytest.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log" | cut -c1-200
t1=$(date +%s); timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? ($(( $(date +%s) - t1 )) s)"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_chunk_latency_ms", "stage_ms")})
r = d["roofline"]; print("roofline:", {k: r.get(k) for k in ("kernel", "frac", "frac_rocprof", "step_frac", "traffic", "avg_us", "rocprof_avg_us")}, r["traffic_detail"]["calibration"]["bytes_per_raw_kib_over_1024"])
c = d["cpu_baseline"]; print("cpu:", c["value"], c["window_s_all_runs"], c["spread"], c["single_thread"]["value"], c["int8"].get("value"))
print("parity:", d["parity_prefix"], "stream p50:", d["stream"]["unpaced"]["p50_chunk_latency_ms"], d["stream"]["paced_256ms"]["p50_chunk_latency_ms"], "throughput:", d["throughput"].get("xrt"))
PY
echo "total $(( $(date +%s) - t0 )) s"
