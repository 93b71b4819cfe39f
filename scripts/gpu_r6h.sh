#!/bin/bash
# Round 6: every GPU test on the tree with the no-slab one-pass prefill, then the conditioned window.
set -u
TAG=${1:-r6h}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -rA > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 "$OUT/pytest.log"
grep -E "^(FAILED|ERROR)|223-step decode|diverges at" "$OUT/pytest.log" | head -20
timeout 200 python scripts/prefill_time.py small.en 2>&1 | grep "small.en" | tee "$OUT/prefill_time.txt"
timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 5 > "$OUT/bench_cond.json" 2> "$OUT/bench_cond.err"
python - "$OUT/bench_cond.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "value_conditioned", "ms_per_step")}, d.get("conditioned_window"))
PY
echo "total $(( $(date +%s) - t0 )) s"
