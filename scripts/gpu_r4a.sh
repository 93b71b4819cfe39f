#!/bin/bash
# round 4, call A: batched decode rows as 16-row tiles (WLX_ROWTILE, decoder.hip) — parity of the batched cases, then the decode step
# per kernel at 60 / 40 / 20 rows (small.en) and 40 rows (large-v3), row tiles on and off
set -u
TAG=r4a; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_batched_depth.py -m gpu -q -x -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
for RT in 1 0; do
  for R in 60 40 20; do
    WLX_ROWTILE=$RT timeout 300 python scripts/step_profile.py small.en $R 33 --json "$OUT/step_small_r${R}_rt$RT.json" 2>&1 | tee -a "$OUT/steps.txt" | head -14
  done
  WLX_ROWTILE=$RT timeout 600 python scripts/step_profile.py large-v3 40 33 --json "$OUT/step_lv3_r40_rt$RT.json" 2>&1 | tee -a "$OUT/steps.txt" | head -16
done
timeout 600 python bench.py --batch 12 --steps 4 --warmup 2 --no-stream --no-cpu-baseline --no-pmc > "$OUT/bench_batch12.json" 2> "$OUT/bench_batch12.err"; tail -c 600 "$OUT/bench_batch12.json" | head -c 300; echo
