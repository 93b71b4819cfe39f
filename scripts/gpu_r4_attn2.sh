#!/bin/bash
# Round 4: XCD-aware workgroup map of the encoder attention (WLX_ENC_ATTN_XCD=0 = plain order) x forms 3 / 6 / 8 — parity, then encode times
set -u
TAG=${1:-r4attn2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
timeout 900 python -m pytest tests/test_gpu_encoder_batched.py tests/test_gpu_parity.py tests/test_gpu_full_depth.py tests/test_jfk_fixture.py -m gpu -q -x -p no:cacheprovider --timeout=800 > "$OUT/pytest3.log" 2>&1; echo "pytest form 3 + map rc=$?"; tail -2 "$OUT/pytest3.log"
WLX_ENC_ATTN=8 timeout 900 python -m pytest tests/test_gpu_encoder_batched.py tests/test_gpu_parity.py tests/test_gpu_full_depth.py -m gpu -q -x -p no:cacheprovider --timeout=800 > "$OUT/pytest8.log" 2>&1; echo "pytest form 8 + map rc=$?"; tail -2 "$OUT/pytest8.log"
enc() { env $1 timeout 300 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s/^/[$1] /" | tee -a "$OUT/encode_ab.txt"; }
for x in 0 1; do for f in 3 6 8; do enc "WLX_ENC_ATTN_XCD=$x WLX_ENC_ATTN=$f" small.en 12; done; done
for x in 0 1; do for f in 3 8; do enc "WLX_ENC_ATTN_XCD=$x WLX_ENC_ATTN=$f" small.en 1; done; done
for x in 0 1; do for f in 3 8; do enc "WLX_ENC_ATTN_XCD=$x WLX_ENC_ATTN=$f" large-v3 1; done; done
for x in 0 1; do for f in 3 8; do enc "WLX_ENC_ATTN_XCD=$x WLX_ENC_ATTN=$f" large-v3 8; done; done
echo done
