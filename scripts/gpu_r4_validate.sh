#!/bin/bash
# Round-4 validation of the tree on one GPU box: every GPU test, smoke(), the default bench line (stream leg, CPU baseline, live
# FETCH_SIZE pass, throughput leg), rocprofv3 kernel stats of the same bench command, and the other configurations.
# usage: scripts/gpu_r4_validate.sh [tag]
set -u
TAG=${1:-r4v}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -rA --durations=15 -p no:cacheprovider --timeout=900 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" "$OUT/pytest_gpu.log" | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log" | cut -c1-200
run_b() {
  name=$1; shift
  timeout 900 python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "p50", round(d.get("p50_chunk_latency_ms", d.get("p50_step_ms", 0)), 2), "cpu", (d.get("cpu_baseline") or {}).get("value"), "parity_prefix", d.get("parity_prefix"))
    if "stage_ms" in d: print("   stage:", {k: round(v, 3) for k, v in d["stage_ms"].items()}, "step", round(d.get("decode_step", {}).get("graph_replay_ms", 0), 4))
    if "roofline" in d: print("   roofline:", {k: v for k, v in d["roofline"].items() if k in ("kernel", "achieved", "frac", "traffic", "avg_us")})
    if "roofline_encoder" in d: print("   roofline_encoder:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline_encoder"].items()})
    if "throughput" in d: print("   throughput:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["throughput"].items() if k != "note"})
    if "conditioned_window" in d: print("   conditioned:", {k: round(v, 2) for k, v in d["conditioned_window"].items() if isinstance(v, float)})
    st = d.get("stream")
    if st: print("   stream:", {k: (round(v["p50_chunk_latency_ms"], 2), round(v["p95_chunk_latency_ms"], 2), round(v["xrt"], 1), v["client_errors"]) for k, v in st.items() if isinstance(v, dict) and "xrt" in v}, st.get("vad", {}).get("audio_kept_fraction"))
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
run_b default
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-stream --no-throughput > "$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp "$F" "$OUT/kernel_stats.csv" && head -14 "$F" | cut -c1-160
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
run_b batch12 --batch 12 --steps 4 --warmup 2 --no-stream --no-cpu-baseline
run_b s4_b12 --streams 4 --batch 12 --steps 3 --warmup 1 --no-stream --no-cpu-baseline --no-pmc
run_b s4 --streams 4 --steps 10 --warmup 2 --no-stream --no-cpu-baseline --no-pmc
run_b large_v3 --model large-v3 --steps 5 --warmup 2 --no-stream --no-cpu-baseline --no-pmc
run_b config5_lanes1 --config 5 --lanes 1 --steps 2 --warmup 1
run_b config5 --config 5 --steps 2 --warmup 1
du -sh "$OUT"
