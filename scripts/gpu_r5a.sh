#!/bin/bash
# Round 5, GPU call 1: the lifted row cap (parity + throughput), the two unmeasured / revisited switches, the repaired bench line.
set -u
TAG=${1:-r5a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=300 --durations=15 -rs > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -8 "$OUT/pytest.log"
timeout 120 python -m pytest tests/test_int8_gap.py tests/test_gpu_lean_family.py -m gpu -q -p no:cacheprovider -s -k "int8 or twentyfour" 2>&1 | grep -E "small.en peaked|logits n=120|passed|failed" > "$OUT/pytest_new_prints.log"; cat "$OUT/pytest_new_prints.log"
B="python bench.py --no-stream --no-cpu-baseline --no-pmc"
short() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("  (no JSON line:", e, ")"); sys.exit(0)
o = {k: d.get(k) for k in ("value", "ms_per_step")}
o["stage"] = d.get("stage_ms"); ds = d.get("decode_step", {})
o["step_rows"] = ds.get("rows"); o["step_ms"] = ds.get("graph_replay_ms")
if "conditioned_window" in d: o["cond_ms"] = d["conditioned_window"]["ms_per_window"]
if "throughput" in d: o["throughput"] = {k: d["throughput"].get(k) for k in ("xrt", "streams", "batch_per_stream", "ms_per_step", "encode_ms_one_slot", "decode_step_ms", "error")}
print(" ", json.dumps(o))
PY
}
echo "== default (full line)"; timeout 700 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"; short "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  roofline:", json.dumps({k: v for k, v in d["roofline"].items() if k not in ("largest_launch",)}))
print("  cpu:", json.dumps(d.get("cpu_baseline"))[:1500]); print("  parity:", d.get("parity_prefix"), " stream:", json.dumps(d.get("stream", {}).get("unpaced")))
PY
for sw in "WLX_NO_EMBED_FOLD=1" "WLX_PREFILL_LN=1"; do
  echo "== $sw"; env $sw timeout 300 $B --no-throughput --steps 10 > "$OUT/bench_${sw%%=*}.json" 2> "$OUT/bench_${sw%%=*}.err"; echo "rc=$?"; short "$OUT/bench_${sw%%=*}.json"
done
for b in 12 24 48; do
  echo "== small.en --batch $b (one slot)"; timeout 400 $B --no-throughput --batch $b --steps 3 --warmup 1 > "$OUT/bench_batch$b.json" 2> "$OUT/bench_batch$b.err"; echo "rc=$?"; short "$OUT/bench_batch$b.json"; tail -2 "$OUT/bench_batch$b.err"
done
for shp in 2x24 4x24 1x48; do
  echo "== throughput shape $shp"; timeout 400 $B --steps 2 --warmup 1 --throughput-shape $shp > "$OUT/bench_tp_$shp.json" 2> "$OUT/bench_tp_$shp.err"; echo "rc=$?"; short "$OUT/bench_tp_$shp.json"
done
for mb in 8 16 32; do
  echo "== config 5 (large-v3, 64 clips) max-batch $mb"; timeout 600 python bench.py --config 5 --no-pmc --steps 2 --warmup 1 --max-batch $mb > "$OUT/bench_c5_mb$mb.json" 2> "$OUT/bench_c5_mb$mb.err"; echo "rc=$?"
  python - "$OUT/bench_c5_mb$mb.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("  ", json.dumps({k: d.get(k) for k in ("value", "ms_per_step", "p50_step_ms")}), json.dumps({k: v for k, v in d.get("decode_step", {}).items() if k != "kernels"}))
except Exception as e:
    print("  (no JSON line:", e, ")")
PY
  tail -2 "$OUT/bench_c5_mb$mb.err"
done
echo "total $(( $(date +%s) - t0 )) s"; du -sh "$OUT"
