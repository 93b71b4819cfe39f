#!/bin/bash
# Round 4: batched residual epilogue (gemm.hip gemm_epilogue_m RESID) — parity, encode times, then the full validation
set -u
TAG=${1:-r4resid}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
timeout 900 python -m pytest tests/test_gpu_encoder_batched.py tests/test_gpu_parity.py tests/test_gpu_full_depth.py tests/test_jfk_fixture.py -m gpu -q -x -p no:cacheprovider --timeout=800 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
enc() { env $1 timeout 300 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s/^/[$1] /" | tee -a "$OUT/encode_ab.txt"; }
enc A=1 small.en 12
enc A=1 small.en 1
enc A=1 large-v3 1
enc A=1 large-v3 8
echo done
