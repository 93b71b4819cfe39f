#!/bin/bash
# Round 5, GPU call 2: the fused self-attention + output projection (dec_sao_kernel), the cleaned-up binary, d_model 384 on the lean kernels.
set -u
TAG=${1:-r5b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=300 --durations=8 -rs > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -6 "$OUT/pytest.log"; grep -E "^FAILED|^ERROR" "$OUT/pytest.log" | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log" | cut -c1-400
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
echo "== default shapes, SAO on (production library)"; timeout 300 $B --no-throughput --steps 20 > "$OUT/bench_sao.json" 2> "$OUT/bench_sao.err"; echo "rc=$?"; short "$OUT/bench_sao.json"
echo "== WLX_NO_SAO=1 (libwlx_ab.so)"; WLX_LIB=$REPO/whisperlive_amd/libwlx_ab.so WLX_NO_SAO=1 timeout 300 $B --no-throughput --steps 20 > "$OUT/bench_nosao.json" 2> "$OUT/bench_nosao.err"; echo "rc=$?"; short "$OUT/bench_nosao.json"
echo "== small.en --batch 12 (gemm3 after the probe removal)"; timeout 400 $B --no-throughput --batch 12 --steps 3 --warmup 1 > "$OUT/bench_batch12.json" 2> "$OUT/bench_batch12.err"; echo "rc=$?"; short "$OUT/bench_batch12.json" | head -1
for shp in 4x32 4x48 3x48; do
  echo "== throughput shape $shp"; timeout 500 $B --steps 2 --warmup 1 --throughput-shape $shp > "$OUT/bench_tp_$shp.json" 2> "$OUT/bench_tp_$shp.err"; echo "rc=$?"; short "$OUT/bench_tp_$shp.json" | head -1; tail -1 "$OUT/bench_tp_$shp.err"
done
echo "== tiny.en single stream"; timeout 300 $B --no-throughput --model tiny.en --steps 10 > "$OUT/bench_tiny.json" 2> "$OUT/bench_tiny.err"; echo "rc=$?"; short "$OUT/bench_tiny.json"
echo "total $(( $(date +%s) - t0 )) s"; du -sh "$OUT"
