#!/bin/bash
set -u
TAG=${1:-r2y}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_depth.py tests/test_gpu_lean_family.py -m gpu -q -x -p no:cacheprovider --timeout=600 > "$OUT/pytest_sub.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_sub.log"
for cfg in "X=0" "WLX_GEMM_EPI_LDS=0" "X=1" "WLX_GEMM_EPI_LDS=0"; do
  echo -n "[$cfg] "; env $cfg timeout 120 python scripts/encode_only.py small.en 6 2>&1 | grep encode_ms
done
echo -n "[large-v3] "; timeout 300 python scripts/encode_only.py large-v3 3 2>&1 | grep encode_ms
echo -n "[large-v3 direct] "; WLX_GEMM_EPI_LDS=0 timeout 300 python scripts/encode_only.py large-v3 3 2>&1 | grep encode_ms
