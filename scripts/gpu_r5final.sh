#!/bin/bash
# Round-5 end-of-round validation: every GPU test, smoke(), the default bench line, rocprofv3 kernel stats of the same command, the other
# configurations, and the counter passes (MFMA ops of the encoder; L2 hit / miss + FETCH_SIZE of the 60-row decode step).
set -u
TAG=${1:-r5final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=300 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 "$OUT/pytest.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log" | cut -c1-300
timeout 700 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"; tail -3 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_chunk_latency_ms", "stage_ms")})
print("roofline:", json.dumps({k: v for k, v in d["roofline"].items() if k != "largest_launch"}))
print("encoder:", d["roofline_encoder"]); print("cond:", d.get("conditioned_window"))
print("cpu:", json.dumps(d.get("cpu_baseline"))[:1200]); print("parity:", d.get("parity_prefix")); print("stream:", json.dumps(d.get("stream", {}).get("unpaced")), json.dumps(d.get("stream", {}).get("paced_256ms")))
print("throughput:", json.dumps(d.get("throughput")))
PY
cp gpurun_out/bench_kernel_stats.csv "$OUT/bench_child_kernel_stats.csv" 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > "$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp "$F" "$OUT/kernel_stats.csv" && head -16 "$F" | cut -c1-170
B="python bench.py --no-stream --no-cpu-baseline --no-throughput"
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ds = d.get("decode_step", {}); r = d.get("roofline", {})
    print("  ", json.dumps({"value": d.get("value"), "ms_per_step": d.get("ms_per_step"), "stage": d.get("stage_ms"), "step_rows": ds.get("rows"), "step_ms": ds.get("graph_replay_ms"),
                             "hbm_frac": ds.get("hbm_frac_of_peak"), "dom": r.get("kernel"), "frac": r.get("frac"), "traffic": r.get("traffic"), "alg": r.get("algorithmic_bytes_per_launch"),
                             "enc_frac": (d.get("roofline_encoder") or {}).get("frac_of_mfma_peak")}))
except Exception as e:
    print("   (no JSON line:", e, ")")
PY
}
echo "== batch 12"; timeout 400 $B --batch 12 --steps 3 --warmup 1 > "$OUT/bench_batch12.json" 2> "$OUT/bench_batch12.err"; line "$OUT/bench_batch12.json"
echo "== batch 24"; timeout 400 $B --batch 24 --steps 3 --warmup 1 > "$OUT/bench_batch24.json" 2> "$OUT/bench_batch24.err"; line "$OUT/bench_batch24.json"
echo "== 4 streams"; timeout 400 $B --no-pmc --streams 4 --steps 10 > "$OUT/bench_s4.json" 2> "$OUT/bench_s4.err"; line "$OUT/bench_s4.json"
echo "== large-v3"; timeout 500 $B --no-pmc --model large-v3 --steps 5 --warmup 2 > "$OUT/bench_large_v3.json" 2> "$OUT/bench_large_v3.err"; line "$OUT/bench_large_v3.json"
echo "== tiny.en"; timeout 300 $B --no-pmc --model tiny.en --steps 10 > "$OUT/bench_tiny_en.json" 2> "$OUT/bench_tiny_en.err"; line "$OUT/bench_tiny_en.json"
for mb in 8 16; do
  echo "== config 5 max-batch $mb"; timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --max-batch $mb > "$OUT/bench_config5_mb$mb.json" 2> "$OUT/bench_config5_mb$mb.err"; line "$OUT/bench_config5_mb$mb.json"
done
cd /tmp
echo "== PMC: MFMA ops, encoder, 1 window and 12 windows"
for b in 1 12; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_mfma_b$b" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 $b > "$OUT/pmc_mfma_b$b.log" 2>&1; echo "rc=$?"
  python "$REPO/scripts/pmc_summary.py" "$OUT/pmc_mfma_b$b" 2>/dev/null | grep -E "gemm3|gemm2|attn_encoder|layernorm" > "$OUT/pmc_mfma_encoder_b$b.csv"; cut -c1-160 "$OUT/pmc_mfma_encoder_b$b.csv" | head -24
done
echo "== PMC: L2 hit / miss + FETCH_SIZE, 60-row decode step (12 windows x 5 beams)"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum FETCH_SIZE --kernel-trace -d "$OUT/pmc_tcc_60rows" -o wlx --output-format csv -- python "$REPO/bench.py" --pmc-child --batch 12 > "$OUT/pmc_tcc.log" 2>&1; echo "rc=$?"
python "$REPO/scripts/pmc_summary.py" "$OUT/pmc_tcc_60rows" 2>/dev/null | grep -E "dec_|search" > "$OUT/pmc_tcc_decode_60rows.csv"; cut -c1-170 "$OUT/pmc_tcc_decode_60rows.csv" | head -45
cd "$REPO"
find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*.db' -delete
echo "total $(( $(date +%s) - t0 )) s"; du -sh "$OUT"
