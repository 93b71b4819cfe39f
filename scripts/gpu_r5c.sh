#!/bin/bash
# Round 5, GPU call 3: tree after the SAO revert + K-split last layer (vocabulary projection reads slabs); full parity suite.
set -u
TAG=${1:-r5c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=300 --durations=5 -rs > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -5 "$OUT/pytest.log"; grep -E "^FAILED|^ERROR" "$OUT/pytest.log" | head -20
B="python bench.py --no-stream --no-cpu-baseline --no-pmc"
short() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("  (no JSON line:", e, ")"); sys.exit(0)
o = {k: d.get(k) for k in ("value", "ms_per_step")}
o["stage"] = d.get("stage_ms"); ds = d.get("decode_step", {})
o["step_rows"] = ds.get("rows"); o["step_ms"] = ds.get("graph_replay_ms"); o["sum_kernel_us"] = ds.get("sum_kernel_us")
if "conditioned_window" in d: o["cond_ms"] = d["conditioned_window"]["ms_per_window"]
if "throughput" in d: o["throughput"] = {k: d["throughput"].get(k) for k in ("xrt", "streams", "batch_per_stream", "ms_per_step", "encode_ms_one_slot", "decode_step_ms", "error")}
print(" ", json.dumps(o))
for k in ds.get("kernels", []): print("     %-46s n=%3d avg %6.2f us" % (k["name"], k["launches"], k["avg_us"]))
PY
}
echo "== small.en single stream"; timeout 300 $B --no-throughput --steps 20 > "$OUT/bench_small.json" 2> "$OUT/bench_small.err"; echo "rc=$?"; short "$OUT/bench_small.json"
echo "== large-v3 single stream"; timeout 400 $B --no-throughput --model large-v3 --steps 5 --warmup 2 > "$OUT/bench_large_v3.json" 2> "$OUT/bench_large_v3.err"; echo "rc=$?"; short "$OUT/bench_large_v3.json"
echo "total $(( $(date +%s) - t0 )) s"
