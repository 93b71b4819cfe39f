#!/bin/bash
# Round 6: four concurrent streams on one GPU (independent slots, per-client decodes) by slot-stream mode: all-CU hardware queues (default), disjoint
# quarters of the CUs (stride4 / contig4), ordinary shared-pool streams (off).
set -u
TAG=${1:-r6l}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp WLX_QUIET=1
for m in full stride4 contig4 off; do for fr in "" "--free-run"; do
  echo "== WLX_SLOT_CU_MASK=$m $fr"; WLX_SLOT_CU_MASK=$m timeout 300 python bench.py --streams 4 $fr --no-stream --no-cpu-baseline --no-throughput --no-pmc --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', round(d['value'],1), 'xRT  ms_per_step', round(d['ms_per_step'],2), 'p50', round(d['p50_chunk_latency_ms'],2), d['stage_ms'])"
done; done 2>&1 | tee "$OUT/streams4_by_slot_stream_mode.txt"
