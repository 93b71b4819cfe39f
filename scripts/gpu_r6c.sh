#!/bin/bash
# Round 6, third GPU call: wide passes (LayerNorm launch + fp16-rows-in projections on 64-row tiles) — parity tests, then A/B against the 16-row tiles
# (WLX_WIDE_MIN_ROWS=0 = off) at 24 / 48 windows per decode (small.en), config 5 at 16 / 32 clips per decode (large-v3) and the conditioned window.
set -u
TAG=${1:-r6c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_batched_depth.py tests/test_gpu_lean_family.py tests/test_gpu_long_context.py tests/test_gpu_ring.py -m gpu -q -p no:cacheprovider --timeout=600 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)" "$OUT/pytest.log" | head -20
B="python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ds = d.get("decode_step", {})
    print("  ", json.dumps({"value": round(d.get("value"), 1), "ms_per_step": round(d.get("ms_per_step"), 2), "stage": d.get("stage_ms"), "step_rows": ds.get("rows"), "step_ms": ds.get("graph_replay_ms"),
                             "cond": (d.get("conditioned_window") or {}).get("ms_per_window")}))
    ks = sorted(ds.get("kernels", []), key=lambda k: -k["total_us"])[:8]
    for k in ks: print("      %-62s n=%5.1f avg %7.2f tot %8.1f" % (k["name"][:62], k["launches"], k["avg_us"], k["total_us"]))
except Exception as e:
    print("   (no JSON line:", e, ")")
PY
}
for w in 64 0; do
  export WLX_WIDE_MIN_ROWS=$w
  echo "=== WLX_WIDE_MIN_ROWS=$w"
  echo "== small.en conditioned window"; timeout 300 $B --steps 5 > "$OUT/bench_cond_w$w.json" 2> "$OUT/bench_cond_w$w.err"; line "$OUT/bench_cond_w$w.json"
  echo "== small.en batch 24"; timeout 300 $B --batch 24 --steps 3 --warmup 1 > "$OUT/bench_b24_w$w.json" 2> "$OUT/bench_b24_w$w.err"; line "$OUT/bench_b24_w$w.json"
  echo "== small.en batch 48"; timeout 300 $B --batch 48 --steps 2 --warmup 1 > "$OUT/bench_b48_w$w.json" 2> "$OUT/bench_b48_w$w.err"; line "$OUT/bench_b48_w$w.json"
  for mb in 16 32; do
    echo "== config 5 max-batch $mb"; timeout 500 python bench.py --config 5 --steps 1 --warmup 1 --max-batch $mb --no-pmc > "$OUT/bench_c5_mb${mb}_w$w.json" 2> "$OUT/bench_c5_mb${mb}_w$w.err"; line "$OUT/bench_c5_mb${mb}_w$w.json"
  done
done
echo "total $(( $(date +%s) - t0 )) s"
