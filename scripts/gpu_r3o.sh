#!/bin/bash
set -u
TAG=r3o; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_long_context.py -m gpu -q -s -p no:cacheprovider --timeout=600 -k "fallback or three_items or options or prompt_forms or one_pass" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^small.en|Error|assert" "$OUT/pytest.log" | cut -c1-220 | tail -30
