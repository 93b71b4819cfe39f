#!/bin/bash
# round 3, call T: concurrency x batching — S slots on their own queues, each decoding B windows per step
set -u
TAG=r3t; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for cfg in "2 6" "2 12" "4 3" "4 6" "3 12"; do
  set -- $cfg
  timeout 600 python bench.py --streams $1 --batch $2 --steps 4 --warmup 2 --no-stream --no-cpu-baseline --no-pmc > "$OUT/bench_s$1_b$2.json" 2> "$OUT/bench_s$1_b$2.err"
  python - "$OUT/bench_s$1_b$2.json" $1 $2 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("streams", sys.argv[2], "x batch", sys.argv[3], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2))
except Exception as e: print("streams", sys.argv[2], "batch", sys.argv[3], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
done
