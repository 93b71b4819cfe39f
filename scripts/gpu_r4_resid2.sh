#!/bin/bash
# Round 4: batched residual epilogue A/B on ONE box (libwlx_resid0.so = -DWLX_RESID_BATCHED=0), interleaved repeats
set -u
TAG=${1:-r4resid2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
enc() { env $1 timeout 300 python scripts/encode_only.py $2 3 $3 2>&1 | grep encode_ms | sed "s|^|[$1] |" | tee -a "$OUT/encode_ab.txt"; }
for rep in 1 2; do
  enc A=batched small.en 12
  enc WLX_LIB=$PWD/whisperlive_amd/libwlx_resid0.so small.en 12
  enc "WLX_ENC_ATTN=3 A=batched" small.en 12
  enc "WLX_ENC_ATTN=3 WLX_LIB=$PWD/whisperlive_amd/libwlx_resid0.so" small.en 12
done
enc A=batched large-v3 8
enc WLX_LIB=$PWD/whisperlive_amd/libwlx_resid0.so large-v3 8
enc A=batched small.en 1
enc WLX_LIB=$PWD/whisperlive_amd/libwlx_resid0.so small.en 1
echo done
