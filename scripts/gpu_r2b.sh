#!/bin/bash
# Round-2 second GPU pass: new encoder kernels (LDS-DMA ring GEMM, prefetching attention) and activation-first decode GEMVs.
# Parity first, then A/B timings against the round-1 forms (env toggles / libwlx_wfirst.so), rocprofv3 stats of the encoder.
set -u
TAG=${1:-r2b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" "$OUT/pytest.log" | tail -3
grep -E "12-layer|32-layer|common prefix|^FAILED|^E  " "$OUT/pytest.log" | head -40
echo "--- encoder A/B (ms per 30 s window, small.en)"
for cfg in "" "WLX_ENC_GEMM=1" "WLX_ENC_ATTN=1" "WLX_ENC_GEMM=1 WLX_ENC_ATTN=1"; do
  echo -n "[$cfg] "; env $cfg timeout 120 python scripts/encode_only.py small.en 6 2>&1 | tail -1
done
echo -n "[large-v3 new] "; timeout 300 python scripts/encode_only.py large-v3 3 2>&1 | tail -1
echo -n "[large-v3 old] "; WLX_ENC_GEMM=1 WLX_ENC_ATTN=1 timeout 300 python scripts/encode_only.py large-v3 3 2>&1 | tail -1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof_enc" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 4 > "$OUT/rocprof_enc.log" 2>&1; echo "rocprof enc rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof_enc" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && head -16 "$F"
find "$OUT" -name '*kernel_trace.csv' -size +5M -delete
echo "--- decode A/B: activation-first (libwlx.so) vs weights-first (libwlx_wfirst.so)"
for lib in "" "WLX_LIB=$REPO/whisperlive_amd/libwlx_wfirst.so"; do
  env $lib timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stream --no-pmc > "$OUT/bench_ab.json" 2> "$OUT/bench_ab.err"
  python - "$OUT/bench_ab.json" "[$lib]" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "stage_ms")}, "step graph ms", d["decode_step"]["graph_replay_ms"])
    for k in d["decode_step"]["kernels"]: print("     ", k["name"], k["launches"], round(k["avg_us"], 2))
except Exception as e:
    print(sys.argv[2], "parse failed", e)
PY
done
echo "--- default bench line"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; grep "\[bench" "$OUT/bench.err" | tail -12
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "p50_chunk_latency_ms", "stage_ms", "parity_prefix", "n_gpus")})
    print("roofline:", d["roofline"]); print("cpu:", d.get("cpu_baseline")); print("stream:", d.get("stream"))
except Exception as e:
    print("bench parse failed", e)
PY
timeout 600 python bench.py --config 5 --steps 2 --warmup 1 > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"; echo "config5 rc=$?"; tail -2 "$OUT/bench_config5.err"
cat "$OUT/bench_config5.json"
du -sh "$OUT"
