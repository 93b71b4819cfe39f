#!/bin/bash
# round 3, call H: one-pass prefill on large-v3 after the two-tile fallback; parity of every module the change touches
set -u
TAG=r3h; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for o in 1 0; do echo "== large-v3 WLX_PREFILL_ONE_PASS=$o"; WLX_PREFILL_ONE_PASS=$o timeout 300 python scripts/prefill_time.py large-v3 2>&1 | grep -E "prompt|prefill"; done > "$OUT/prefill_time_large_v3.txt" 2>&1; cat "$OUT/prefill_time_large_v3.txt"
timeout 1200 python -m pytest tests/test_gpu_long_context.py tests/test_gpu_lean_family.py tests/test_gpu_batched_depth.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
timeout 600 python bench.py --model large-v3 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-stream > "$OUT/bench_large_v3.json" 2> "$OUT/bench_large_v3.err"; echo "bench rc=$?"
python - "$OUT/bench_large_v3.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print({k: d[k] for k in ("value", "ms_per_step")}, d.get("conditioned_window"), d["stage_ms"], d["decode_step"]["graph_replay_ms"])
PY
du -sh "$OUT"
