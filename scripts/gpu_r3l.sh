#!/bin/bash
set -u
TAG=r3l; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for f in 0 1; do echo "== small.en WLX_NO_FUSED_CQ=$f"; WLX_NO_FUSED_CQ=$f timeout 300 python scripts/prefill_time.py small.en 2>&1 | grep -E "prompt|prefill"; done > "$OUT/prefill_fused_cq.txt" 2>&1; cat "$OUT/prefill_fused_cq.txt"
