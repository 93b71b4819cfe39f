#!/bin/bash
# Round-2 decode pass 2: K-split MLP output projection (KS 2 / 4), embedding fold, symmetric-wave slab prologue.
set -u
TAG=${1:-r2g}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
for lib in "" "WLX_LIB=$REPO/whisperlive_amd/libwlx_ks4.so"; do
  env $lib timeout 600 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_full_depth.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=600 > "$OUT/pytest_sub.log" 2>&1; echo "[$lib] pytest rc=$?"
  tail -3 "$OUT/pytest_sub.log"
done
timeout 200 python scripts/trace_step.py --csv "$OUT/trace.csv" > "$OUT/trace.txt" 2>&1; echo "trace rc=$?"
head -12 "$OUT/trace.txt"; tail -6 "$OUT/trace.txt"
for cfg in "X=0" "WLX_XS_WAVES=8" "WLX_LIB=$REPO/whisperlive_amd/libwlx_ks4.so" "WLX_LIB=$REPO/whisperlive_amd/libwlx_ks4.so WLX_XS_WAVES=8" "WLX_FC2_KS=0 WLX_NO_EMBED_FOLD=1"; do
  env $cfg timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-stream --no-pmc > "$OUT/bench_quick.json" 2> "$OUT/bench_quick.err"
  python - "$OUT/bench_quick.json" "[$cfg]" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "stage_ms", "parity_prefix")}, "step graph ms", d["decode_step"]["graph_replay_ms"])
    for k in d["decode_step"]["kernels"]: print("     ", k["name"], k["launches"], round(k["avg_us"], 2))
except Exception as e:
    print(sys.argv[2], "parse failed", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
