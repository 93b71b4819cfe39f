#!/bin/bash
# Round-2 re-entry baseline: fresh decode-step trace (in-kernel timeline) + quick bench of the committed tree.
set -u
TAG=${1:-r2e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 200 python scripts/trace_step.py --csv "$OUT/trace.csv" > "$OUT/trace.txt" 2>&1; echo "trace rc=$?"
head -24 "$OUT/trace.txt"; tail -6 "$OUT/trace.txt"
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stream --no-pmc > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"
python - "$OUT/bench_quick.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "stage_ms")}, "step graph ms", d["decode_step"]["graph_replay_ms"])
for k in d["decode_step"]["kernels"]: print("     ", k["name"], k["launches"], round(k["avg_us"], 2))
PY
