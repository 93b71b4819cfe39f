#!/bin/bash
# Log G5: which tests differ with WLX_G2_CHMAX=12 on the d_model 1280 family, and how (first failures in full).
set -u
TAG=${1:-r6ap}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_LIB=whisperlive_amd/libwlx_ab.so
WLX_G2_CHMAX=12 timeout 900 python -m pytest tests/test_gpu_lean_family.py -m gpu -q -p no:cacheprovider --timeout=900 --tb=short 2>&1 | tail -150 > "$OUT/pytest_lean_family_chmax12.txt"
WLX_G2_CHMAX=12 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 --tb=line 2>&1 | grep -E "FAILED|passed|failed|Error" | head -60 > "$OUT/pytest_all_failed_chmax12.txt"
tail -3 "$OUT/pytest_all_failed_chmax12.txt"
