#!/bin/bash
# Round 6: how much of a one-window encoder GEMM launch is cold operands? Every gemm2 launch issued twice (libwlx_probe_twice.so): first against second.
set -u
export TMPDIR=/tmp WLX_QUIET=1; REPO=$PWD; OUT=$PWD/gpurun_out/${1:-r6y}; mkdir -p $OUT; cd /tmp
for m in small.en; do
  WLX_LIB=$REPO/whisperlive_amd/libwlx_probe_twice.so timeout 300 rocprofv3 --kernel-trace -d "$OUT/p" -o enc --output-format csv -- python $REPO/scripts/encode_only.py $m 4 1 2>/dev/null | tail -1
  python $REPO/scripts/trace_alternate.py "$OUT/p"; rm -rf "$OUT/p"
done 2>&1 | tee $OUT/gemm2_first_vs_second_launch.txt
