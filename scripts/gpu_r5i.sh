#!/bin/bash
# Round 5: row tiles of two / three MFMA row tiles for the residual projections of wide batches — parity and the batched legs.
set -u
TAG=${1:-r5i}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
timeout 600 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_batched_depth.py tests/test_session_and_batch.py tests/test_gpu_transcriber.py -m gpu -q -p no:cacheprovider --timeout=300 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
B="python bench.py --no-stream --no-cpu-baseline --no-pmc"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); ds = d.get("decode_step", {})
    o = {"value": d.get("value"), "ms_per_step": d.get("ms_per_step"), "step_rows": ds.get("rows"), "step_ms": ds.get("graph_replay_ms")}
    if "throughput" in d: o["throughput"] = {k: d["throughput"].get(k) for k in ("xrt", "streams", "batch_per_stream", "ms_per_step", "decode_step_ms")}
    print("  ", json.dumps(o))
except Exception as e:
    print("   (no JSON line:", e, ")")
PY
}
for b in 24 48; do echo "== batch $b"; timeout 400 $B --no-throughput --batch $b --steps 3 --warmup 1 > "$OUT/bench_batch$b.json" 2> "$OUT/bench_batch$b.err"; line "$OUT/bench_batch$b.json"; done
echo "== throughput 3x48"; timeout 500 $B --steps 2 --warmup 1 > "$OUT/bench_tp.json" 2> "$OUT/bench_tp.err"; line "$OUT/bench_tp.json"
for mb in 16 32; do echo "== config 5 max-batch $mb"; timeout 600 python bench.py --config 5 --no-pmc --steps 2 --warmup 1 --max-batch $mb > "$OUT/bench_c5_mb$mb.json" 2> "$OUT/bench_c5_mb$mb.err"; line "$OUT/bench_c5_mb$mb.json"; done
