#!/bin/bash
# Round 4: do any HIP runtime knobs move the captured decode step (launch-chain bound)? step graph time, one stream (5 rows) and 60 rows
set -u
TAG=${1:-r4env}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
run() { env $1 timeout 300 python scripts/step_profile.py small.en $2 33 2>&1 | grep "^==" | sed "s/^==/== [$1]/" | tee -a "$OUT/knobs.txt"; }
for k in A=1 HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 ROC_USE_FGS_KERNARG=0 DEBUG_HIP_KERNARG_COPY_OPT=0 AMD_OPT_FLUSH=0 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1024 ROC_ACTIVE_WAIT_TIMEOUT=1000 A=2; do
  run $k 5
done
run A=1 60
run HIP_FORCE_DEV_KERNARG=0 60
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 60
echo done
