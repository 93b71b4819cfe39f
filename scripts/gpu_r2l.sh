#!/bin/bash
set -u
TAG=${1:-r2l}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 300 python scripts/trace_step.py --model large-v3 --items 8 --t 20 --csv "$OUT/trace_l40.csv" > "$OUT/trace_l40.txt" 2>&1; echo "trace rc=$?"
head -20 "$OUT/trace_l40.txt"; tail -5 "$OUT/trace_l40.txt"
timeout 300 python scripts/trace_step.py --model small.en --items 8 --t 20 --csv "$OUT/trace_s40.csv" > "$OUT/trace_s40.txt" 2>&1; echo "trace rc=$?"
head -12 "$OUT/trace_s40.txt"; tail -5 "$OUT/trace_s40.txt"
