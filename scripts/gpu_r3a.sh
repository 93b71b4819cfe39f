#!/bin/bash
# round 3, call A: the new parity tests (long context, batched depth, checkpoint route, jfk) + the 4-stream overlap diagnosis
set -u
TAG=r3a; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_long_context.py tests/test_jfk_fixture.py tests/test_gpu_checkpoint_route.py tests/test_gpu_lean_family.py \
   -m gpu -q -s -rA -p no:cacheprovider --timeout=900 > "$OUT/pytest_new.log" 2>&1; echo "pytest new rc=$?"
grep -E "passed|failed|error" "$OUT/pytest_new.log" | tail -3
timeout 1500 python -m pytest tests/test_gpu_batched_depth.py -m gpu -q -s -rA -p no:cacheprovider --timeout=1200 > "$OUT/pytest_lv3.log" 2>&1; echo "pytest lv3 rc=$?"
grep -E "passed|failed|error" "$OUT/pytest_lv3.log" | tail -3
# ---- 4 concurrent slots: throughput under three launch regimes, then the kernel timeline of the default one
for cfg in "default" "GPU_MAX_HW_QUEUES=8" "WLX_NO_GRAPH=1" "GPU_MAX_HW_QUEUES=8 WLX_NO_GRAPH=1"; do
  if [ "$cfg" = default ]; then envs=""; else envs="$cfg"; fi
  env $envs timeout 300 python bench.py --streams 4 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-stream > "$OUT/streams4_${cfg// /_}.json" 2> "$OUT/streams4.err"
  python - "$OUT/streams4_${cfg// /_}.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("streams4", sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "p50", round(d["p50_chunk_latency_ms"], 2))
except Exception as e: print("streams4", sys.argv[2], "FAILED", e)
PY
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/rocprof_s4" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --streams 4 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-stream > "$OUT/rocprof_s4.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof_s4" -name '*kernel_trace.csv' | head -1)
[ -n "$F" ] && python scripts/stream_overlap.py "$F" > "$OUT/streams4_overlap.txt" 2>&1 && python scripts/stream_overlap.py "$F" decode > "$OUT/streams4_overlap_decode.txt" 2>&1
cat "$OUT/streams4_overlap_decode.txt"
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
du -sh "$OUT"
