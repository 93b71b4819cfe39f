#!/bin/bash
set -u
TAG=r3v; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python bench.py --config 5 --max-batch 12 --clips 96 --steps 2 --warmup 1 --no-pmc > "$OUT/bench_config5_mb12.json" 2> "$OUT/bench_config5_mb12.err"
python - "$OUT/bench_config5_mb12.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("config5 max_batch 12, 96 clips: xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), d.get("decode_step", {}).get("graph_replay_ms"))
except Exception as e: print("FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
