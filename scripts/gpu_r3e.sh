#!/bin/bash
# round 3, call E: the whole GPU suite on the current tree (dedicated-queue slot streams by default), smoke, the default bench line,
# configs[2] with out-of-process clients, 8 slots on one GPU (fallback to the shared pool)
set -u
TAG=r3e; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA --durations=25 -p no:cacheprovider --timeout=600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" "$OUT/pytest_gpu.log" | tail -3; grep -E "s call|s setup" "$OUT/pytest_gpu.log" | head -14
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log" | cut -c1-300
run_b() {
  name=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 900 python bench.py "${args[@]}" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "p50", round(d.get("p50_chunk_latency_ms", d.get("p50_step_ms", 0)), 2), "cpu", (d.get("cpu_baseline") or {}).get("value"), "parity_prefix", d.get("parity_prefix"))
    st = d.get("stream")
    if st: print("   stream:", {k: (round(v["p50_chunk_latency_ms"], 2), round(v["p95_chunk_latency_ms"], 2), round(v["xrt"], 1), v["client_errors"]) for k, v in st.items() if isinstance(v, dict) and "xrt" in v}, st.get("vad"))
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run_b default -- A=1
run_b small_4clients --model small --stream-clients 4 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc -- A=1
run_b small_4clients_batch --model small --stream-clients 4 --stream-batch --steps 3 --warmup 1 --no-cpu-baseline --no-pmc -- A=1
run_b s8 --streams 8 --steps 6 --warmup 2 --no-stream --no-cpu-baseline --no-pmc -- A=1
du -sh "$OUT"
