#!/bin/bash
# Round 5, GPU call 5: per-GEMM tile shapes of the single-window encoder (policy vs the 64 x 96 tile everywhere), encoder parity, headline.
set -u
TAG=${1:-r5e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_full_depth.py tests/test_gpu_encoder_batched.py tests/test_gpu_parity.py tests/test_gpu_lean_family.py tests/test_jfk_fixture.py -m gpu -q -p no:cacheprovider --timeout=300 -k "encoder or encode or full_depth or logmel or jfk or batched" > "$OUT/pytest_encoder.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_encoder.log"
cd /tmp
table() { python - "$1" "$2" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "prep_window" in r["Kernel_Name"])
seg = rows[idx:]
agg = {}
out = []
for r in seg:
    nm = r["Kernel_Name"].replace("void wlx::", "").replace("(wlx::GemmParams)", "")
    if "layernorm" in nm: nm = "layernorm_kernel"
    if "attn_encoder" in nm: nm = "attn_encoder_lds_kernel"
    if "prep_window" in nm: nm = "prep_window_kernel"
    key = (nm[:28], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.append("%-30s grid %8s x %-4s  n=%3d  avg %8.2f us  total %8.1f us" % (k[0], k[1], k[2], v[0], v[1] / v[0], v[1]))
out.append("launches %d, sum %.1f us" % (len(seg), tot))
open(sys.argv[2], "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
}
for m in small.en large-v3; do
  for mode in policy shape0; do
    rm -rf /tmp/kt; if [ $mode = policy ]; then EV=""; else EV="WLX_LIB=$REPO/whisperlive_amd/libwlx_ab.so WLX_GEMM2_SHAPE=0"; fi
    env $EV timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o wlx --output-format csv -- python $REPO/scripts/encode_only.py $m 4 1 > "$OUT/enc_${m}_$mode.log" 2>&1
    echo "== $m $mode: $(tail -1 "$OUT/enc_${m}_$mode.log" | grep -o 'encode_ms.*')"; table /tmp/kt "$OUT/encoder_launches_${m}_$mode.txt"
  done
done
cd $REPO
echo "== headline"; timeout 300 python bench.py --no-stream --no-cpu-baseline --no-pmc --no-throughput --steps 20 > "$OUT/bench_small.json" 2> "$OUT/bench_small.err"; python - "$OUT/bench_small.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print({k: d[k] for k in ("value", "ms_per_step", "stage_ms")}, d["roofline_encoder"])
PY
echo "total $(( $(date +%s) - t0 )) s"
