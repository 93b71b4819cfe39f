#!/bin/bash
set -u
TAG=${1:-r4u}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
timeout 1200 python -m pytest tests/test_gpu_lean_family.py -m gpu -q -x -p no:cacheprovider --timeout=1100 -k "busy or twelve" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
