#!/bin/bash
set -u
TAG=${1:-r2h}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
timeout 600 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_full_depth.py -m gpu -q -x -p no:cacheprovider --timeout=600 > "$OUT/pytest_sub.log" 2>&1; echo "pytest rc=$?"
tail -3 "$OUT/pytest_sub.log"
timeout 200 python scripts/trace_step.py --csv "$OUT/trace.csv" > "$OUT/trace.txt" 2>&1; echo "trace rc=$?"
head -10 "$OUT/trace.txt"; tail -4 "$OUT/trace.txt"
for cfg in "X=0" "WLX_SELF_ATTN_IDENT=0" "X=1" "WLX_SELF_ATTN_IDENT=0"; do
  env $cfg timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stream --no-pmc > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"
  python - "$OUT/bench_quick.json" "[$cfg]" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "stage_ms", "parity_prefix")}, "step graph ms", d["decode_step"]["graph_replay_ms"])
    for k in d["decode_step"]["kernels"]: print("     ", k["name"], k["launches"], round(k["avg_us"], 2))
except Exception as e:
    print(sys.argv[2], "parse failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
