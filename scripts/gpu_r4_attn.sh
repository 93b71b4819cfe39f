#!/bin/bash
# Round 4: software-pipelined encoder attention (WLX_ENC_ATTN=6/7/8) — parity with the form forced, then encode times
set -u
TAG=${1:-r4attn}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
WLX_ENC_ATTN=6 timeout 900 python -m pytest tests/test_gpu_encoder_batched.py tests/test_gpu_parity.py tests/test_gpu_full_depth.py tests/test_jfk_fixture.py -m gpu -q -x -p no:cacheprovider --timeout=800 > "$OUT/pytest6.log" 2>&1; echo "pytest form 6 rc=$?"; tail -2 "$OUT/pytest6.log"
enc() { env $1 timeout 300 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s/^/[$1] /" | tee -a "$OUT/encode_ab.txt"; }
for f in 3 6 7 8; do enc WLX_ENC_ATTN=$f small.en 12; done
for f in 3 6 7 8; do enc WLX_ENC_ATTN=$f small.en 1; done
for f in 3 6 7; do enc WLX_ENC_ATTN=$f large-v3 1; done
for f in 3 6; do enc WLX_ENC_ATTN=$f large-v3 8; done
echo done
