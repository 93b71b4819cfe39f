#!/bin/bash
set -u
TAG=r3q; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 300 python scripts/step_by_position.py small.en 2>&1 | grep "decode step" > "$OUT/step_by_position.txt"; timeout 300 python scripts/step_by_position.py large-v3 2>&1 | grep "decode step" >> "$OUT/step_by_position.txt"; cat "$OUT/step_by_position.txt"
timeout 900 python -m pytest tests/test_gpu_long_context.py tests/test_gpu_parity.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-stream > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print({k: d[k] for k in ("value", "ms_per_step")}, d["decode_step"]["graph_replay_ms"], d.get("conditioned_window"))
PY
