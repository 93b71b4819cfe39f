#!/bin/bash
# round 3, call B: 4-slot concurrency under queue / CU-mask regimes, the new bench legs, timing of the new parity modules
set -u
TAG=r3b; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
run_s4() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --streams 4 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-stream > "$OUT/streams4_$name.json" 2> "$OUT/streams4_$name.err"
  python - "$OUT/streams4_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("streams4", sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "p50", round(d["p50_chunk_latency_ms"], 2))
except Exception as e: print("streams4", sys.argv[2], "FAILED", e)
PY
}
run_s4 default A=1
run_s4 q5 GPU_MAX_HW_QUEUES=5
run_s4 q6 GPU_MAX_HW_QUEUES=6
run_s4 cu_full WLX_SLOT_CU_MASK=full
run_s4 cu_contig4 WLX_SLOT_CU_MASK=contig4
run_s4 cu_stride4 WLX_SLOT_CU_MASK=stride4
run_s4 cu_full_q8 WLX_SLOT_CU_MASK=full GPU_MAX_HW_QUEUES=8
cd /tmp
WLX_SLOT_CU_MASK=full timeout 600 rocprofv3 --kernel-trace -d "$OUT/rocprof_s4" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --streams 4 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-stream > "$OUT/rocprof_s4.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof_s4" -name '*kernel_trace.csv' | head -1)
[ -n "$F" ] && python scripts/stream_overlap.py "$F" decode > "$OUT/streams4_cu_full_overlap_decode.txt" 2>&1
cat "$OUT/streams4_cu_full_overlap_decode.txt"
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
# ---- the new bench legs
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}); print("roofline", {k: v for k, v in d["roofline"].items() if k != "largest_launch"}); print("stream", d.get("stream"))
PY
timeout 900 python bench.py --config 5 --clips 16 --steps 1 --warmup 1 > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"; echo "config5 rc=$?"
python - "$OUT/bench_config5.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}); print("roofline", d.get("roofline")); print("decode_step", {k: v for k, v in d.get("decode_step", {}).items() if k != "kernels"})
PY
timeout 900 python bench.py --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --no-stream > "$OUT/bench_batch4.json" 2> "$OUT/bench_batch4.err"; echo "batch4 rc=$?"
tail -c 600 "$OUT/bench_batch4.err"
# ---- timing of the new parity modules
timeout 1200 python -m pytest tests/test_gpu_long_context.py tests/test_gpu_batched_depth.py -m gpu -q -s -rA --durations=20 -p no:cacheprovider --timeout=900 > "$OUT/pytest_new.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error|s call|s setup" "$OUT/pytest_new.log" | tail -30
du -sh "$OUT"
