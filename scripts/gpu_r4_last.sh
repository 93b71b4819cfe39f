#!/bin/bash
# Round 4: last check of the tree — encoder-side parity tests, smoke, batch-12 and default bench lines
set -u
TAG=${1:-r4last}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder_batched.py tests/test_gpu_parity.py tests/test_gpu_full_depth.py tests/test_jfk_fixture.py tests/test_trained_tiny.py tests/test_gpu_batched_depth.py tests/test_gpu_transcriber.py -m gpu -q -p no:cacheprovider --timeout=800 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log" | cut -c1-160
for cfg in "batch12 --batch 12 --steps 6 --warmup 2" "s4_b12 --streams 4 --batch 12 --steps 4 --warmup 1" "default --steps 20 --warmup 3 --no-throughput"; do
  set -- $cfg; name=$1; shift
  timeout 600 python bench.py "$@" --no-stream --no-cpu-baseline --no-pmc > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), {k: round(v, 3) for k, v in d.get("stage_ms", {}).items()}, "roofline_encoder", round(d["roofline_encoder"]["frac_of_mfma_peak"], 4))
except Exception as e: print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
echo done
