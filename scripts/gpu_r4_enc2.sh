#!/bin/bash
# Round 4: batched prep_window (parity), residual GEMMs on the second form (A/B), gemm3 on fewer CUs under four slots (A/B)
set -u
TAG=${1:-r4enc2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
timeout 900 python -m pytest tests/test_gpu_encoder_batched.py tests/test_gpu_parity.py tests/test_gpu_batched_depth.py tests/test_jfk_fixture.py -m gpu -q -x -p no:cacheprovider --timeout=800 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
enc() { env $1 timeout 300 python scripts/encode_only.py small.en 3 12 2>&1 | grep encode_ms | sed "s/^/[$1] /" | tee -a "$OUT/encode_ab.txt"; }
enc A=1
enc "WLX_GEMM3_RESID=0 WLX_GEMM2_LARGE_SHAPE=7"
enc "WLX_GEMM3_RESID=0 WLX_GEMM2_LARGE_SHAPE=0"
enc "WLX_GEMM3_RESID=0"
run() { env $1 timeout 900 python bench.py $2 --no-stream --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$1] $2', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d.get('stage_ms',{}).items()})" | tee -a "$OUT/bench_ab.txt"; }
run A=1 "--streams 4 --batch 12 --steps 4 --warmup 1"
run WLX_GEMM3_CUS=224 "--streams 4 --batch 12 --steps 4 --warmup 1"
run WLX_GEMM3_CUS=192 "--streams 4 --batch 12 --steps 4 --warmup 1"
run A=1 "--batch 12 --steps 6 --warmup 2"
echo done
