#!/bin/bash
# Rehearsal of bench.py's N-rank code path on a one-GPU box (WLX_BENCH_REHEARSAL=1: ranks share the GPU, gloo instead of RCCL): does every rank
# get through the barriers, the reductions and the final line? Not a measurement.
set -u
TAG=${1:-r5v}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1 WLX_BENCH_REHEARSAL=1
t0=$(date +%s)
timeout 400 python bench.py --gpus 2 --steps 4 --warmup 1 > "$OUT/rehearsal_2ranks.json" 2> "$OUT/rehearsal_2ranks.err"; echo "2 ranks via respawn rc=$? ($(( $(date +%s) - t0 )) s)"; tail -c 600 "$OUT/rehearsal_2ranks.json"; echo; tail -5 "$OUT/rehearsal_2ranks.err" | cut -c1-300
t1=$(date +%s)
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 3 --warmup 1 > "$OUT/rehearsal_4ranks.json" 2> "$OUT/rehearsal_4ranks.err"; echo "4 ranks, the driver's launch line rc=$? ($(( $(date +%s) - t1 )) s)"; tail -c 400 "$OUT/rehearsal_4ranks.json"; echo; tail -5 "$OUT/rehearsal_4ranks.err" | cut -c1-300
t2=$(date +%s)
timeout 500 python bench.py --gpus 2 --config 5 --steps 1 --warmup 1 --clips 16 --max-batch 8 > "$OUT/rehearsal_config5.json" 2> "$OUT/rehearsal_config5.err"; echo "config 5, 2 ranks rc=$? ($(( $(date +%s) - t2 )) s)"; tail -c 400 "$OUT/rehearsal_config5.json"; echo; tail -5 "$OUT/rehearsal_config5.err" | cut -c1-300
