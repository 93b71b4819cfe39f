#!/bin/bash
# round 3, call G: one-pass prompt prefill — time by form, parity, the conditioned window in the bench line
set -u
TAG=r3g; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for m in small.en large-v3; do
  for o in 1 0; do echo "== $m WLX_PREFILL_ONE_PASS=$o"; WLX_PREFILL_ONE_PASS=$o timeout 300 python scripts/prefill_time.py $m 2>&1 | grep -E "prompt|prefill"; done
done > "$OUT/prefill_time.txt" 2>&1; cat "$OUT/prefill_time.txt"
timeout 900 python -m pytest tests/test_gpu_long_context.py tests/test_gpu_parity.py tests/test_gpu_transcriber.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-stream > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print({k: d[k] for k in ("value", "ms_per_step")}, d.get("conditioned_window"))
PY
WLX_PREFILL_ONE_PASS=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-stream > "$OUT/bench_chunked.json" 2> "$OUT/bench_chunked.err"
python - "$OUT/bench_chunked.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("chunked prefill:", d.get("conditioned_window"))
PY
du -sh "$OUT"
