#!/bin/bash
# Round 6: the attention output projection and the fused cross-attention kernel as ONE launch with the dependency carried inside it (WLX_FUSE_OC=1):
# parity, then A/B of the headline and the decode step; in-kernel timeline of the fused step.
set -u
TAG=${1:-r6m}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
WLX_FUSE_OC=1 timeout 900 python -m pytest tests/test_gpu_full_depth.py tests/test_gpu_long_context.py tests/test_gpu_lean_family.py tests/test_trained_tiny.py -m gpu -q -p no:cacheprovider --timeout=600 -x > "$OUT/pytest_fused.log" 2>&1; echo "pytest (fused) rc=$? ($(( $(date +%s) - t0 )) s)"; tail -3 "$OUT/pytest_fused.log"
for v in 1 0 1 0; do
  echo "== WLX_FUSE_OC=$v"; WLX_FUSE_OC=$v timeout 300 python bench.py --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 10 --warmup 3 2>/dev/null > "$OUT/bench_oc$v.json"
  python - "$OUT/bench_oc$v.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ds = d["decode_step"]
print("  ", round(d["value"], 1), "xRT", round(d["ms_per_step"], 3), "ms; step graph", round(1e3 * ds["graph_replay_ms"], 1), "us; generate", round(d["stage_ms"]["generate_ms"], 3))
for k in sorted(ds["kernels"], key=lambda k: -k["total_us"])[:6]: print("      %-58s n=%5.1f avg %7.2f" % (k["name"][:58], k["launches"], k["avg_us"]))
PY
done
WLX_FUSE_OC=1 WLX_LIB=whisperlive_amd/libwlx_trace.so timeout 300 python scripts/trace_step.py --model small.en --t 33 > "$OUT/decode_step_trace_fused.txt" 2>&1; sed -n 1,12p "$OUT/decode_step_trace_fused.txt"; tail -2 "$OUT/decode_step_trace_fused.txt"
echo "total $(( $(date +%s) - t0 )) s"
