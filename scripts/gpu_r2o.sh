#!/bin/bash
set -u
TAG=${1:-r2o}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
for f in 3 4 5; do
WLX_ENC_ATTN=$f timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider --timeout=600 -k "encoder" > "$OUT/pytest_enc.log" 2>&1; echo "form $f pytest rc=$?"; tail -1 "$OUT/pytest_enc.log"
done
for cfg in "WLX_ENC_ATTN=3" "WLX_ENC_ATTN=4" "WLX_ENC_ATTN=5" "WLX_ENC_ATTN=2" "WLX_ENC_ATTN=3" "WLX_ENC_ATTN=4" "WLX_ENC_ATTN=5"; do
  echo -n "[$cfg] "; env $cfg timeout 120 python scripts/encode_only.py small.en 6 2>&1 | grep encode_ms
done
