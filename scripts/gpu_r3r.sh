#!/bin/bash
set -u
TAG=r3r; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_trained_tiny.py -m gpu -q -s -p no:cacheprovider --timeout=300 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|word error|Error|assert" "$OUT/pytest.log" | cut -c1-260 | tail -12
