#!/bin/bash
# config 5 (large-v3, 64 clips, batches of 8) kernel breakdown under rocprofv3 --stats, and the large-v3 single-stream line
set -u
TAG=${1:-r2i}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof_c5" -o wlx --output-format csv -- python "$REPO/bench.py" --config 5 --clips 16 --steps 1 --warmup 1 > "$OUT/c5.json" 2> "$OUT/c5.err"; echo "rocprof c5 rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof_c5" -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && head -30 "$F" | cut -c1-200
find "$OUT" -name '*kernel_trace.csv' -size +3M -delete; find "$OUT" -name '*agent_info.csv' -delete
tail -1 "$OUT/c5.json" | cut -c1-600
timeout 600 python bench.py --model large-v3 --steps 6 --warmup 2 --no-cpu-baseline --no-stream --no-pmc > "$OUT/bench_large.json" 2> "$OUT/bench_large.err"
python - "$OUT/bench_large.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "stage_ms")}, "step graph ms", d["decode_step"]["graph_replay_ms"])
for k in d["decode_step"]["kernels"]: print("     ", k["name"], k["launches"], round(k["avg_us"], 2), k["bytes_per_launch"])
PY
