#!/bin/bash
set -u
TAG=r3p; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_long_context.py tests/test_gpu_transcriber.py tests/test_gpu_batched_depth.py -m gpu -q -s -p no:cacheprovider --timeout=600 -k "short_prompts or three_items or batch or batched" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^item|Error|assert" "$OUT/pytest.log" | cut -c1-220 | tail -20
for j in 1 0; do
  WLX_PREFILL_JOINT=$j timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-pmc > "$OUT/bench_config5_joint$j.json" 2> "$OUT/bench_config5_joint$j.err"
  python - "$OUT/bench_config5_joint$j.json" $j <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("config5 joint", sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1))
except Exception as e: print("config5 joint", sys.argv[2], "FAILED", e)
PY
done
