#!/bin/bash
# Round 6: where the 223-row prompt prefill goes (rocprofv3 kernel trace of conditioned windows, grouped by kernel and grid).
set -u
TAG=${1:-r6e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1; REPO=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d "$OUT/prof" -o wlx --output-format csv -- python "$REPO/scripts/prefill_trace.py" small.en 4 3 > "$OUT/prefill_trace.log" 2>&1; echo "rc=$?"; tail -3 "$OUT/prefill_trace.log"
cd "$REPO"
F=$(find "$OUT/prof" -name '*kernel_trace.csv' | head -1)
python - "$F" > "$OUT/prefill_by_kernel_grid.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST pass: find the last dec_embed_kernel / first kernel with grid rows >= 200 ... simply aggregate launches whose Grid_Size_X suggests the prefill (exclude decode steps by taking kernels between the last encoder kernel and the first search kernel)
idx_enc = max(i for i, r in enumerate(rows) if "attn_encoder" in r["Kernel_Name"] or "gemm2_kernel" in r["Kernel_Name"])
idx_srch = min(i for i, r in enumerate(rows) if i > idx_enc and "search_scan3" in r["Kernel_Name"])
seg = rows[idx_enc + 1: idx_srch]
# the prefill = everything up to the first vocab kernel of the first decode step; split at the first dec_vocab_kernel with small grid... keep all, report order
agg = collections.OrderedDict()
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
for r in seg:
    k = (r["Kernel_Name"].split("(")[0][-70:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
print("segment (after the encoder, before the first search): %d launches, %.1f us wall, %.1f us of kernels" % (len(seg), (t1 - t0) / 1e3, sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s grid %6s x %3s x %3s wg %4s  n=%3d avg %7.2f us total %8.1f us" % (k[0], k[1], k[2], k[3], k[4], v[0], v[1] / v[0], v[1]))
PY
cat "$OUT/prefill_by_kernel_grid.txt"
find "$OUT/prof" -name '*.csv' -delete; find "$OUT/prof" -name '*.db' -delete
