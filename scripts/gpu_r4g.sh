#!/bin/bash
# round 4, call G: where gemm3's time goes — per-launch durations by GEMM shape, SQ wait / MFMA-busy / LDS-conflict counters, L2 hit rate
set -u
TAG=${1:-r4g}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
for CFG in "1 small.en 12" "1 large-v3 8"; do
  set -- $CFG; D="$OUT/rp_${2%%.*}_$3"
  WLX_GEMM3=$1 timeout 600 rocprofv3 --kernel-trace --stats -d "$D" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" $2 2 $3 > "$D.log" 2>&1; echo "rocprof rc=$?"
  python "$REPO/scripts/trace_by_grid.py" "$D" "$2 B=$3 GEMM3=$1" | tee -a "$OUT/gemm_launches.txt"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace \
  -d "$OUT/pmc_sq" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 12 > "$OUT/pmc_sq.log" 2>&1; echo "pmc sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace \
  -d "$OUT/pmc_mfma" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 12 > "$OUT/pmc_mfma.log" 2>&1; echo "pmc mfma rc=$?"
cd "$REPO"
python scripts/pmc_summary.py "$OUT/pmc_sq" 2>/dev/null | grep -E "gemm3|gemm2|attn_encoder" > "$OUT/pmc_sq_summary.csv"; cut -c1-150 "$OUT/pmc_sq_summary.csv"
python scripts/pmc_summary.py "$OUT/pmc_mfma" 2>/dev/null | grep -E "gemm3|gemm2|attn_encoder" > "$OUT/pmc_mfma_summary.csv"; cut -c1-150 "$OUT/pmc_mfma_summary.csv"
find "$OUT" -name '*.csv' -size +1M -delete
