#!/bin/bash
# Round-6 end-of-round validation on the final tree: every GPU test, smoke(), the default bench line exactly as the driver runs it, rocprofv3
# --kernel-trace --stats of the same command (the bench's own child profilers off: no nested profilers), and the decode-only child the roofline's
# rocprof average comes from (bench.py keeps gpurun_out/bench_kernel_stats.csv).
set -u
TAG=${1:-r6final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -4 "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)|223-step decode|diverges at" "$OUT/pytest.log" | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log" | cut -c1-200
t1=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? ($(( $(date +%s) - t1 )) s)"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "value_conditioned", "ms_per_step", "p50_chunk_latency_ms", "stage_ms", "h2d_excluded_ms")})
r = d["roofline"]; print("roofline:", {k: r.get(k) for k in ("kernel", "frac", "frac_rocprof", "step_frac", "traffic", "avg_us", "rocprof_avg_us")}, r["traffic_detail"]["calibration"]["bytes_per_raw_kib_over_1024"])
c = d["cpu_baseline"]; print("cpu:", c["value"], c["window_s_all_runs"], c["spread"], c["single_thread"]["value"], c["int8"].get("value"))
print("parity:", d["parity_prefix"], "stream p50:", d["stream"]["unpaced"]["p50_chunk_latency_ms"], d["stream"]["paced_256ms"]["p50_chunk_latency_ms"], d["stream"].get("stage_ms_per_chunk"), "throughput:", d["throughput"].get("xrt"))
PY
cp gpurun_out/bench_kernel_stats.csv "$OUT/kernel_stats_decode_child.csv" 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > "$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
f=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_bench_command.csv" && head -12 "$f" | cut -c1-160
find "$OUT/rocprof" -name '*kernel_trace.csv' -delete; find "$OUT/rocprof" -name '*.db' -delete
# counter passes, each in its own run (kernel-trace only beside --pmc): MFMA counters of the single-window encoder; L2 hit / miss of the single-stream decode
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_mfma_b1" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 1 > "$OUT/pmc_mfma_b1.log" 2>&1; echo "pmc mfma rc=$?"
python "$REPO/scripts/pmc_summary.py" "$OUT/pmc_mfma_b1" 2>/dev/null | grep -E "gemm3|gemm2|attn_encoder|layernorm" > "$OUT/pmc_mfma_encoder_b1.csv"; cut -c1-150 "$OUT/pmc_mfma_encoder_b1.csv" | head -24
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OUT/pmc_tcc" -o wlx --output-format csv -- python "$REPO/bench.py" --pmc-child > "$OUT/pmc_tcc.log" 2>&1; echo "pmc tcc rc=$?"
python "$REPO/scripts/pmc_summary.py" "$OUT/pmc_tcc" 2>/dev/null | grep -E "dec_|search" > "$OUT/pmc_tcc_decode.csv"; cut -c1-150 "$OUT/pmc_tcc_decode.csv" | head -30
cd "$REPO"
find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*.db' -delete
echo "total $(( $(date +%s) - t0 )) s"; du -sh "$OUT"
