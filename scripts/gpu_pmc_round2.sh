#!/bin/bash
# Counter passes over the round-2 kernels (each its own run, kernel-trace only): FETCH_SIZE of one window's launches (decode
# step + encoder), and the SQ wait/active split of the encoder kernels.  usage: scripts/gpu_pmc_round2.sh <tag>
set -u
TAG=${1:-pmc2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o wlx --output-format csv -- \
  python "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-stream --no-pmc > "$OUT/pmc_fetch.log" 2>&1; echo "pmc fetch rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace \
  -d "$OUT/pmc_sq" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 > "$OUT/pmc_sq.log" 2>&1; echo "pmc sq rc=$?"
cd "$REPO"
python scripts/pmc_summary.py "$OUT/pmc_fetch" > "$OUT/pmc_fetch_summary.csv" 2>&1; grep -E "dec_|gemm2|attn_enc|search" "$OUT/pmc_fetch_summary.csv" | cut -c1-150 | head -30
python scripts/pmc_summary.py "$OUT/pmc_sq" 2>/dev/null | grep -E "gemm2|attn_encoder" > "$OUT/pmc_sq_summary.csv"; cut -c1-140 "$OUT/pmc_sq_summary.csv"
find "$OUT" -name '*counter_collection.csv' -size +3M -delete; find "$OUT" -name '*kernel_trace.csv' -size +3M -delete
