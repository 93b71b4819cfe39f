#!/bin/bash
# Round 4: Infinity-Cache probe for the batched cross attention (scripts/mall_probe.hip) + step A/Bs (embedding fold, fused combine)
set -u
TAG=${1:-r4mall}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
timeout 120 scripts/_build/mall_probe 2>&1 | tee "$OUT/mall_probe.txt"
prof() { env $1 timeout 600 python scripts/step_profile.py $2 $3 33 2>&1 | sed "s/^==/== [$1]/" | tee -a "$OUT/steps.txt" | head -${4:-3}; }
prof A=1 small.en 60 12
prof WLX_EMBED_FOLD_BATCHED=1 small.en 60 4
prof WLX_XATTN_SEPARATE=0 small.en 60 12
prof A=1 large-v3 40 4
prof WLX_XATTN_SEPARATE=0 large-v3 40 4
timeout 900 python -m pytest tests/test_gpu_lean_family.py tests/test_gpu_transcriber.py -m gpu -q -x -p no:cacheprovider --timeout=800 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$OUT/pytest.log"
echo done
