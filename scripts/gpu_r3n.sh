#!/bin/bash
# round 3, call N: 49..64 decoder rows on the lean kernels in row chunks — parity (12 items batched, 60 teacher-forced rows) and
# the aggregate throughput of wider batches
set -u
TAG=r3n; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lean_family.py -m gpu -q -p no:cacheprovider --timeout=600 -k "twelve or eight_items or rows_1_to_48" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
for B in 8 10 12; do
  timeout 600 python bench.py --batch $B --steps 4 --warmup 2 --no-stream --no-cpu-baseline --no-pmc > "$OUT/bench_batch$B.json" 2> "$OUT/bench_batch$B.err"
  python - "$OUT/bench_batch$B.json" $B <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("batch", sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "decode step", round(d["decode_step"]["graph_replay_ms"], 4), d["stage_ms"])
except Exception as e: print("batch", sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
done
