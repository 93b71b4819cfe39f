#!/bin/bash
# Round-2 diagnosis pass: where do the encoder kernels' cycles go? SQ / cache counters (their own rocprofv3 passes), encoder
# A/B against the round-1 forms, GEMM tile-shape sweep.
set -u
TAG=${1:-r2c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
rocprofv3 -L > "$OUT/counters.txt" 2>&1; grep -c . "$OUT/counters.txt"
grep -oE "\b(SQ_[A-Z_0-9]+|TA_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|TCC_[A-Z_0-9a-z]+|TD_[A-Z_0-9a-z]+|GRBM_[A-Z_0-9]+)\b" "$OUT/counters.txt" | sort -u > "$OUT/counter_names.txt"; wc -l "$OUT/counter_names.txt"
grep -E "^(SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_VMEM|SQ_ACTIVE_INST_LDS|SQ_ACTIVE_INST_MISC|SQ_ACTIVE_INST_SCA|SQ_INSTS_VALU|SQ_INSTS_VMEM_RD|SQ_INSTS_LDS|SQ_INSTS_SALU|SQ_INSTS_MFMA|SQ_INSTS_VALU_MFMA_MOPS_F16|SQ_VALU_MFMA_BUSY_CYCLES|SQ_WAIT_INST_LDS|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_INST_CYCLES_VMEM|SQ_WAVES|SQ_IFETCH|TA_BUSY_avr|TA_BUSY_sum|TA_TA_BUSY_sum|TCP_TCC_READ_REQ_sum|TCP_TOTAL_CACHE_ACCESSES_sum|TCP_TA_TCP_STATE_READ_sum|TCP_PENDING_STALL_CYCLES_sum|TCP_TCP_TA_DATA_STALL_CYCLES_sum|TCC_HIT_sum|TCC_MISS_sum|TCC_REQ_sum|TCC_READ_sum|TCC_EA0_RDREQ_sum|TCC_BUSY_sum|TD_TD_BUSY_sum|GRBM_GUI_ACTIVE)$" "$OUT/counter_names.txt" | tr '\n' ' '; echo
pass() {  # name, counters...
  local name=$1; shift
  cd /tmp
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/pmc_$name" -o wlx --output-format csv -- python "$REPO/scripts/encode_only.py" small.en 2 > "$OUT/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"
  cd "$REPO"
  python scripts/pmc_summary.py "$OUT/pmc_$name" 2>/dev/null | grep -E "gemm2|attn_encoder|layernorm|^kernel" | head -60 > "$OUT/pmc_${name}_summary.csv"
  cat "$OUT/pmc_${name}_summary.csv" | cut -c1-170
  find "$OUT/pmc_$name" -name '*.csv' -size +3M -delete
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVES
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
pass mem1 TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass mem2 TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
echo "--- encoder A/B (encode_ms per 30 s window, small.en, 6 encodes)"
for cfg in "X=0" "WLX_ENC_GEMM=1" "WLX_ENC_ATTN=1" "WLX_GEMM2_SHAPE=0" "WLX_GEMM2_SHAPE=3" "WLX_GEMM2_SHAPE=4" "WLX_GEMM2_SHAPE=5" "WLX_GEMM2_SHAPE=6" "WLX_GEMM2_SHAPE=1"; do
  echo -n "[$cfg] "; env $cfg timeout 120 python scripts/encode_only.py small.en 6 2>&1 | grep encode_ms
done
du -sh "$OUT"
