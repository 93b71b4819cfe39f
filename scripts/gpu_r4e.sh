#!/bin/bash
# round 4, call E: gemm3 with the LDS-transposed epilogue — parity, batched / single-window encoder times, per-launch durations
set -u
TAG=${1:-r4e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_encoder_batched.py -m gpu -q -x -s -p no:cacheprovider --timeout=800 > "$OUT/pytest_batched.log" 2>&1; echo "pytest batched rc=$?"; grep -E "passed|failed|max rel|Error|assert" "$OUT/pytest_batched.log" | head -12
WLX_GEMM3=2 timeout 900 python -m pytest tests/test_gpu_full_depth.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=800 -k "encoder" > "$OUT/pytest_forced.log" 2>&1; echo "pytest forced rc=$?"; tail -3 "$OUT/pytest_forced.log"
enc() { env "$1" timeout 600 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s/^/$1 /"; }
{
for B in 12 8 4; do enc WLX_GEMM3=1 small.en $B; done
enc WLX_GEMM_EPI_LDS=0 small.en 12
enc WLX_GEMM3=0 small.en 12
enc WLX_GEMM3=1 large-v3 8
enc WLX_GEMM3=0 large-v3 8
enc WLX_GEMM3=2 small.en 1
enc WLX_GEMM3=0 small.en 1
enc WLX_GEMM3=2 large-v3 1
enc WLX_GEMM3=0 large-v3 1
} | tee "$OUT/encode_times.txt"
cd /tmp
for CFG in "1 small.en 12" "2 small.en 1" "1 large-v3 8"; do
  set -- $CFG
  WLX_GEMM3=$1 timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/rp_${2%%.*}_$3" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" $2 2 $3 > "$OUT/rp_${2%%.*}_$3.log" 2>&1; echo "rocprof rc=$?"
  python - "$OUT/rp_${2%%.*}_$3/wlx_kernel_trace.csv" "$2 B=$3 GEMM3=$1" <<'PY' | tee -a "$OUT/gemm_launches.txt"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
g = [r for r in rows if 'gemm3' in r['Kernel_Name'] or 'gemm2' in r['Kernel_Name'] or 'attn_enc' in r['Kernel_Name'] or 'layernorm' in r['Kernel_Name']]
n = len(g) // 2
acc = collections.OrderedDict()
for r in g[n:]:
    k = (r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = acc.setdefault(k, [0, 0]); a[0] += 1; a[1] += d
print("==", sys.argv[2], "last pass, by (kernel, grid): launches, avg us, total us")
for k, (c, t) in acc.items(): print("  ", k, c, round(t / c / 1e3, 1), round(t / 1e3, 1))
print("   total us", round(sum(t for c, t in acc.values()) / 1e3, 1))
PY
done
find "$OUT" -name '*.csv' -size +1M -delete
