#!/bin/bash
# round 4, call J: what four independent slots contend on (kernel traces of --streams 1 / 4), and the fixed costs of one wlx_generate call
set -u
TAG=${1:-r4j}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
for S in 1 4; do
  timeout 600 rocprofv3 --kernel-trace -d "$OUT/rp_s$S" -o wlx --output-format csv -- python "$REPO/bench.py" --streams $S --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-stream --no-throughput > "$OUT/rp_s$S.log" 2>&1; echo "rocprof s$S rc=$?"
done
cd "$REPO"
python scripts/contention.py "$(find $OUT/rp_s1 -name '*kernel_trace.csv' | head -1)" "$(find $OUT/rp_s4 -name '*kernel_trace.csv' | head -1)" | tee "$OUT/contention.txt"
python scripts/stream_overlap.py "$(find $OUT/rp_s4 -name '*kernel_trace.csv' | head -1)" decode-only > "$OUT/streams4_overlap.txt" 2>&1; tail -12 "$OUT/streams4_overlap.txt"
WLX_GEN_TRACE=1 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-stream --no-throughput 2> "$OUT/gen_trace.txt" > /dev/null; grep -i "gen" "$OUT/gen_trace.txt" | tail -4
find "$OUT" -name '*.csv' -size +1M -delete
