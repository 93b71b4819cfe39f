#!/bin/bash
# Round 6: self-attention K rows fetched like the V rows (lane = (position group, 8-dim chunk): whole 128-byte rows per load instruction) — the decode
# step by position and the prompt prefill, A/B against the lane = position form (libwlx_klane.so = the same tree with the previous kernel), then
# every decode parity test.
set -u
TAG=${1:-r6r}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for m in small.en large-v3; do for lib in libwlx.so libwlx_klane.so libwlx.so libwlx_klane.so; do
  WLX_LIB=whisperlive_amd/$lib timeout 300 python scripts/step_by_position.py $m 2>&1 | tail -1
done; done | tee "$OUT/step_by_position_ab.txt"
for lib in libwlx.so libwlx_klane.so; do echo "== $lib"; WLX_LIB=whisperlive_amd/$lib timeout 300 python scripts/prefill_time.py 2>&1 | grep -i "prefill" | tail -4; done | tee "$OUT/prefill_ab.txt"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -x -k "not ring and not rccl" 2>&1 | tail -5 | tee "$OUT/pytest_tail.txt"
