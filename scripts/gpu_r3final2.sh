#!/bin/bash
# re-validation of the final tree after the last engine changes (slab guard, joint prefill, self-attention prologue)
set -u
TAG=r3final2; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --timeout=600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log" | cut -c1-160
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print({k: d[k] for k in ("value", "ms_per_step", "parity_prefix")}, d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d.get("conditioned_window", {}).get("ms_per_window"), {k: v["p50_chunk_latency_ms"] for k, v in d["stream"].items() if isinstance(v, dict) and "xrt" in v})
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-stream > "$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp "$F" "$OUT/kernel_stats.csv" && head -6 "$F" | cut -c1-150
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
