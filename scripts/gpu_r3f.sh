#!/bin/bash
# round 3, call F: prompt-prefill time by chunk size, long-context parity on the new prefill default, two concurrent streams,
# configs[2] through the batch worker with the adaptive window
set -u
TAG=r3f; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for m in small.en large-v3; do
  for c in 48 64; do echo "== $m WLX_PREFILL_ROWS=$c"; WLX_PREFILL_ROWS=$c timeout 300 python scripts/prefill_time.py $m 2>&1 | grep -v amdgpu.ids; done
done > "$OUT/prefill_time.txt" 2>&1; cat "$OUT/prefill_time.txt"
timeout 900 python -m pytest tests/test_gpu_long_context.py tests/test_gpu_parity.py tests/test_gpu_transcriber.py -m gpu -q -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
run_b() {
  name=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 900 python bench.py "${args[@]}" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "p50", round(d.get("p50_chunk_latency_ms", d.get("p50_step_ms", 0)), 2))
    st = d.get("stream")
    if st: print("   stream:", {k: (round(v["p50_chunk_latency_ms"], 2), round(v["p95_chunk_latency_ms"], 2), round(v["xrt"], 1), v["client_errors"]) for k, v in st.items() if isinstance(v, dict) and "xrt" in v})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run_b s2 --streams 2 --steps 10 --warmup 2 --no-stream --no-cpu-baseline --no-pmc -- A=1
run_b s3 --streams 3 --steps 10 --warmup 2 --no-stream --no-cpu-baseline --no-pmc -- A=1
run_b small_4clients_batch --model small --stream-clients 4 --stream-batch --steps 3 --warmup 1 --no-cpu-baseline --no-pmc -- A=1
du -sh "$OUT"
