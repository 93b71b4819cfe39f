#!/bin/bash
# Round 4, end: per-launch encoder table (12 windows) + the MFMA counter pass of the final tree
set -u
TAG=${1:-r4pmcf}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD; export WLX_QUIET=1
cd /tmp
D="$OUT/rp_small_12"
timeout 600 rocprofv3 --kernel-trace --stats -d "$D" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 12 > "$D.log" 2>&1; echo "rocprof rc=$?"
python "$REPO/scripts/trace_by_grid.py" "$D" "small.en B=12 final tree" | tee "$OUT/gemm_launches.txt"
cp "$D"/*kernel_stats.csv "$OUT/kernel_stats_encoder_b12.csv" 2>/dev/null || find "$D" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_encoder_b12.csv" \;
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace \
  -d "$OUT/pmc_mfma" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 12 > "$OUT/pmc_mfma.log" 2>&1; echo "pmc mfma rc=$?"
cd "$REPO"
python scripts/pmc_summary.py "$OUT/pmc_mfma" 2>/dev/null | grep -E "gemm3|gemm2|attn_encoder" > "$OUT/pmc_mfma_summary.csv"; cut -c1-170 "$OUT/pmc_mfma_summary.csv"
find "$OUT" -name '*.csv' -size +1M -delete
echo done
