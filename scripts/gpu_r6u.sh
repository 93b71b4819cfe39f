export TMPDIR=/tmp WLX_QUIET=1
for m in small.en large-v3; do for lib in libwlx.so libwlx_lnold.so libwlx.so libwlx_lnold.so; do
  echo -n "$m $lib  "; WLX_LIB=whisperlive_amd/$lib timeout 300 python scripts/encode_only.py $m 20 1 2>/dev/null | tail -1
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_encoder_batched.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -2
