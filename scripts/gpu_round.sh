#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, bench line, A/B against the first-generation decode
# kernels, rocprofv3 kernel stats of the same bench command.  usage: scripts/gpu_round.sh <tag> [--skip-tests]
set -u
TAG=${1:-r01}; shift || true
SKIP_TESTS=0; [ "${1:-}" = "--skip-tests" ] && SKIP_TESTS=1
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
if [ $SKIP_TESTS = 0 ]; then
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=240 > "$OUT/pytest.log" 2>&1
  echo "pytest rc=$?"; tail -25 "$OUT/pytest.log"
fi
timeout 600 python bench.py --steps 10 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "stage_ms")})
    print("step:", {k: v for k, v in d["decode_step"].items() if k != "kernels"})
    for k in d["decode_step"]["kernels"]: print("  ", k)
    print("roofline:", d["roofline"]); print("cpu:", d.get("cpu_baseline"))
except Exception as e:
    print("bench parse failed:", e)
PY
tail -5 "$OUT/bench.err"
WLX_DECODE_V1=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_v1.json" 2> "$OUT/bench_v1.err"
echo "bench v1 rc=$?"; python -c "
import json,sys
d=json.loads(open('$OUT/bench_v1.json').read().strip().splitlines()[-1]); print('v1:', d['value'], d['ms_per_step'], d['stage_ms'], d['decode_step']['graph_replay_ms'])" 2>&1 | tail -2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/rocprof.log" 2>&1
echo "rocprof rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && head -30 "$F"
find "$OUT/rocprof" -name '*kernel_trace.csv' -size +20M -delete
du -sh "$OUT"
