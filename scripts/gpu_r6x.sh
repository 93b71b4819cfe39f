#!/bin/bash
# Round 6: encoder attention (one window) — ring depth 3 / 4 / 6 / 8 and the key-split timing probe (-DWLX_PROBE_ATTN_SPLIT: two workgroups per
# query block, half the key tiles each, outputs not merged: timing only), per-launch time from rocprofv3.
set -u
export TMPDIR=/tmp WLX_QUIET=1; REPO=$PWD; OUT=$PWD/gpurun_out/${1:-r6x}; mkdir -p $OUT; cd /tmp
for m in tiny.en small.en large-v3; do for lib in libwlx.so libwlx_ad3.so libwlx_ad6.so libwlx_ad8.so libwlx_probe_split.so; do
  [ -f $REPO/whisperlive_amd/$lib ] || continue
  WLX_LIB=$REPO/whisperlive_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/p" -o enc --output-format csv -- python $REPO/scripts/encode_only.py $m 6 1 2>/dev/null | tail -1
  f=$(find "$OUT/p" -name '*kernel_stats.csv' | head -1); echo "== $m $lib"; grep -E "attn_encoder" "$f" | cut -d, -f2-4,6,7
  rm -rf "$OUT/p"
done; done 2>&1 | tee $OUT/attn_depth_and_split_probe.txt
