#!/bin/bash
# Runs on the GPU box: VAD parity tests, the server loopback e2e test, VAD timing + kernel trace.
TAG=${1:-vad}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python -m pytest tests/test_vad_model.py -m gpu -q -x -p no:cacheprovider --timeout=200 > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -15 $OUT/pytest.log
timeout 120 python scripts/vad_bench.py 30 > $OUT/vad_bench.json 2> $OUT/vad_bench.err; cat $OUT/vad_bench.json; tail -3 $OUT/vad_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o vad --output-format csv -- python $GRAFT_REPO_ROOT/scripts/vad_bench.py 30 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err
cd $GRAFT_REPO_ROOT; f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -8 "$f"
