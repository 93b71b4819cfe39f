#!/bin/bash
set -u
TAG=r3m; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for S in 1 4; do
  WLX_GEN_TRACE=1 timeout 300 python bench.py --streams $S --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-stream > "$OUT/bench_s$S.json" 2> "$OUT/bench_s$S.err"
  echo "== streams $S"; grep "wlx gen" "$OUT/bench_s$S.err" | tail -8
done
