#!/bin/bash
set -u
TAG=r3s; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider --timeout=600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"
