#!/bin/bash
# L2 hit / miss of the 60-row decode step (12 windows x 5 beams): are the row-tile workgroups of a weight tile served by one XCD's L2?
set -u
TAG=${1:-r5g}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; REPO=$PWD; export TMPDIR=/tmp WLX_QUIET=1
cd /tmp
timeout 800 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OUT/pmc_tcc" -o wlx --output-format csv -- python "$REPO/bench.py" --pmc-child --batch 12 > "$OUT/pmc_tcc.log" 2>&1; echo "tcc rc=$?"; tail -2 "$OUT/pmc_tcc.log"
python "$REPO/scripts/pmc_summary.py" "$OUT/pmc_tcc" 2>/dev/null | grep -E "dec_|search" > "$OUT/pmc_tcc_decode_60rows.csv"; cut -c1-170 "$OUT/pmc_tcc_decode_60rows.csv" | head -40
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o wlx --output-format csv -- python "$REPO/bench.py" --pmc-child --batch 12 > "$OUT/pmc_fetch.log" 2>&1; echo "fetch rc=$?"
python "$REPO/scripts/pmc_summary.py" "$OUT/pmc_fetch" 2>/dev/null | grep -E "dec_|search" > "$OUT/pmc_fetch_decode_60rows.csv"; cut -c1-170 "$OUT/pmc_fetch_decode_60rows.csv" | head -24
find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*.db' -delete
