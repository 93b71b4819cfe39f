#!/bin/bash
# Round 4: kernel trace of a conditioned window (225-token prompt prefill + 8 steps)
set -u
TAG=${1:-r4pf}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp; export WLX_QUIET=1
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/rp" -o wlx --output-format csv -- python "$REPO/scripts/prefill_trace.py" small.en 8 3 > "$OUT/rp.log" 2>&1; echo "rocprof rc=$?"; grep pass "$OUT/rp.log"
cd "$REPO"
python scripts/trace_table.py "$OUT/rp" "small.en conditioned window, 8 steps" | tee "$OUT/prefill_table.txt" | head -60
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
echo done
