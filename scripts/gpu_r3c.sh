#!/bin/bash
# round 3, call C: persistent-layer prototype; slot-stream variants (dedicated queue by CU mask / by priority); configs[2] latency;
# the concurrency tests under the new default; timing of the large-v3 parity module with the oracle on the right thread count
set -u
TAG=r3c; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
( cd scripts/ubench && timeout 120 ./persist_layer 12 30 ) > "$OUT/ubench_persist_layer.txt" 2>&1; echo "ubench rc=$?"; cat "$OUT/ubench_persist_layer.txt"
run_b() {  # name, bench args..., -- env...
  name=$1; shift
  args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  env "$@" timeout 400 python bench.py "${args[@]}" --no-cpu-baseline --no-pmc > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "p50", round(d["p50_chunk_latency_ms"], 2))
    st = d.get("stream")
    if st: print("   stream:", {k: (round(v["p50_chunk_latency_ms"], 2), round(v["p95_chunk_latency_ms"], 2), round(v["xrt"], 1)) for k, v in st.items() if isinstance(v, dict) and "xrt" in v})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run_b s1_default --steps 10 --warmup 2 --no-stream -- A=1
run_b s4_default --streams 4 --steps 10 --warmup 2 --no-stream -- A=1
run_b s4_off --streams 4 --steps 10 --warmup 2 --no-stream -- WLX_SLOT_CU_MASK=off
run_b s4_prio_high --streams 4 --steps 10 --warmup 2 --no-stream -- WLX_SLOT_CU_MASK=prio_high
run_b s4_prio_alt --streams 4 --steps 10 --warmup 2 --no-stream -- WLX_SLOT_CU_MASK=prio_alt
run_b s8_default --streams 8 --steps 6 --warmup 2 --no-stream -- A=1
run_b small_4clients --model small --stream-clients 4 --steps 3 --warmup 1 -- A=1
run_b small_4clients_off --model small --stream-clients 4 --steps 3 --warmup 1 -- WLX_SLOT_CU_MASK=off
run_b small_4clients_prio --model small --stream-clients 4 --steps 3 --warmup 1 -- WLX_SLOT_CU_MASK=prio_high
run_b small_4clients_batch --model small --stream-clients 4 --stream-batch --steps 3 --warmup 1 -- A=1
timeout 900 python -m pytest tests/test_gpu_transcriber.py tests/test_server.py tests/test_gpu_batched_depth.py -m gpu -q -rA --durations=12 -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error|s call|s setup" "$OUT/pytest.log" | tail -20
du -sh "$OUT"
