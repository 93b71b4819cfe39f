#!/bin/bash
set -u
TAG=${1:-r4z2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
t() { env $1 timeout 1200 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_encoder_batched.py -m gpu -q -x -p no:cacheprovider --timeout=1100 -k "twelve or eight_items or busy or batched_encoder_equals or rows_1_to_48" > "$OUT/pytest_$2.log" 2>&1; echo "$1 rc=$? $(grep -E 'passed|failed' $OUT/pytest_$2.log | tail -1)"; }
t WLX_ROWTILE=0 rowtile0
t WLX_ROWTILE_CHUNK=32 chunk32
t WLX_ROWTILE_CHUNK=48 chunk48
t A=1 default
