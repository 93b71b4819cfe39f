#!/bin/bash
# Round-end validation on the GPU box: every GPU test, smoke(), the default bench line (with the stream leg and the CPU
# baseline), and the rocprofv3 kernel stats of the same bench command.  usage: scripts/gpu_final.sh <tag>
set -u
TAG=${1:-final}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=240 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_chunk_latency_ms", "stage_ms")})
print("roofline:", {k: v for k, v in d["roofline"].items() if k != "largest_launch"})
print("cpu:", d.get("cpu_baseline")); print("stream:", d.get("stream"))
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/rocprof.log" 2>&1; echo "rocprof rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && head -24 "$F"
find "$OUT" -name '*kernel_trace.csv' -size +5M -delete
du -sh "$OUT"
