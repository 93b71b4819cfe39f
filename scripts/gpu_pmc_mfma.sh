#!/bin/bash
# MFMA counters of the encoder kernels (their own --pmc pass, kernel-trace only).  usage: scripts/gpu_pmc_mfma.sh <tag>
TAG=${1:-pmc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
  -d "$OUT/pmc_mfma" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 3 > "$OUT/pmc_mfma.log" 2>&1
echo "pmc rc=$?"; tail -3 "$OUT/pmc_mfma.log"
cd "$REPO"
python scripts/pmc_summary.py "$OUT/pmc_mfma" > "$OUT/pmc_mfma_summary.csv" 2>&1; head -40 "$OUT/pmc_mfma_summary.csv"
F=$(find "$OUT/pmc_mfma" -name '*kernel_trace.csv' | head -1)
[ -n "$F" ] && python - "$F" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel,launches,avg_ns_under_pmc")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f'"{k}",{len(v)},{sum(v)/len(v):.0f}')
PY
find "$OUT" -name '*counter_collection.csv' -size +5M -delete
