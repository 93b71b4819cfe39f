#!/bin/bash
# Round 4: row tiles for the wide LayerNorm projections by default — parity (large-v3 40 rows, family), steps at 20 / 40 / 60 rows, config 5
set -u
TAG=${1:-r4lv3rt2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
timeout 1200 python -m pytest tests/test_gpu_batched_depth.py tests/test_gpu_lean_family.py tests/test_gpu_full_depth.py -m gpu -q -p no:cacheprovider --timeout=1100 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
prof() { env $1 timeout 600 python scripts/step_profile.py $2 $3 33 2>&1 | sed "s/^==/== [$1]/" | tee -a "$OUT/steps.txt" | head -${4:-1}; }
prof A=1 large-v3 20
prof WLX_ROWTILE_WIDE=0 large-v3 20
prof A=1 large-v3 60
prof WLX_ROWTILE_WIDE=0 large-v3 60
prof A=1 medium 40
prof WLX_ROWTILE_WIDE=0 medium 40
run() { env $1 timeout 900 python bench.py $2 --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$1] $2', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a "$OUT/bench_ab.txt"; }
run A=1 "--config 5 --lanes 1 --steps 2 --warmup 1"
run A=1 "--config 5 --steps 2 --warmup 1"
echo done
