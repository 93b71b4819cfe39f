#!/bin/bash
# round 4, call Z: the documented A/B switches still give correct results (parity subset under each)
set -u
TAG=${1:-r4z}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
t() { env $1 timeout 1200 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_encoder_batched.py -m gpu -q -x -p no:cacheprovider --timeout=1100 -k "twelve or eight_items or busy or batched_encoder_equals or rows_1_to_48" > "$OUT/pytest_$2.log" 2>&1; echo "$1 rc=$? $(tail -1 $OUT/pytest_$2.log)"; }
t WLX_ROWTILE=0 rowtile0
t WLX_VOCAB2=0 vocab0
t WLX_VOCAB2=1 vocab1
t WLX_GEMM3=0 gemm3off
t WLX_RT_NTB2=0 ntb2off
t WLX_RT_NTB4=0 ntb4off
t WLX_RT_F16_NTB2=1 f16ntb2
t WLX_FC2_KS_BATCHED=1 ksbatched
t WLX_ROWTILE_CHUNK=32 chunk32
t WLX_GEMM_EPI_LDS=0 epilds0
t WLX_SLOT_CU_MASK=off cumaskoff
