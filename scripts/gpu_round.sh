#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, bench line, rocprofv3 kernel stats of the same bench command,
# the FETCH_SIZE PMC pass (its own run), and the batched / large-v3 bench lines.  usage: scripts/gpu_round.sh <tag> [--skip-tests]
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
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rocprof" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/rocprof.log" 2>&1
echo "rocprof rc=$?"
cd "$REPO"
F=$(find "$OUT/rocprof" -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && head -30 "$F"
# HBM traffic counters: their own pass (no --stats / trace domains besides kernel-trace), one window only
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
cd "$REPO"
python scripts/pmc_summary.py "$OUT/pmc_fetch" > "$OUT/pmc_fetch_summary.csv" 2>&1; head -30 "$OUT/pmc_fetch_summary.csv"
find "$OUT" -name '*kernel_trace.csv' -size +5M -delete
find "$OUT" -name '*counter_collection.csv' -size +5M -delete
for B in 4 6; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --batch $B > "$OUT/bench_batch$B.json" 2> "$OUT/bench_batch$B.err"
  python -c "import json; d=json.loads(open('$OUT/bench_batch$B.json').read().strip().splitlines()[-1]); print('batch', $B, 'xRT', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2))" || tail -3 "$OUT/bench_batch$B.err"
done
timeout 500 python bench.py --model large-v3 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_large_v3.json" 2> "$OUT/bench_large_v3.err"
python -c "import json; d=json.loads(open('$OUT/bench_large_v3.json').read().strip().splitlines()[-1]); print('large-v3 xRT', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), d['stage_ms'], 'step graph ms', d['decode_step']['graph_replay_ms'])" || tail -3 "$OUT/bench_large_v3.err"
du -sh "$OUT"
