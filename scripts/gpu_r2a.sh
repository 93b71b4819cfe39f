#!/bin/bash
# Round-2 first GPU pass: every GPU test (incl. the full-depth parity tests), the default bench line (live PMC traffic,
# parity_prefix, CPU baseline at nproc + 1 thread), config 5 at N=1.   usage: scripts/gpu_r2a.sh <tag>
set -u
TAG=${1:-r2a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -s > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" "$OUT/pytest.log" | tail -5
grep -E "12-layer|32-layer|common prefix|AssertionError|^FAILED|^E  " "$OUT/pytest.log" | head -40
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -3 "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "p50_chunk_latency_ms", "stage_ms", "parity_prefix", "n_gpus")})
print("step:", {k: v for k, v in d["decode_step"].items() if k != "kernels"})
for k in d["decode_step"]["kernels"]: print("  ", k)
print("roofline:", {k: v for k, v in d["roofline"].items()})
print("cpu:", d.get("cpu_baseline")); print("parity:", d.get("parity")); print("stream:", d.get("stream"))
PY
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 > "$OUT/bench_config5.json" 2> "$OUT/bench_config5.err"; echo "config5 rc=$?"; tail -3 "$OUT/bench_config5.err"
cat "$OUT/bench_config5.json"
du -sh "$OUT"
