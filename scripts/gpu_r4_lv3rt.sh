#!/bin/bash
# Round 4: large-v3 40-row step — row tiles for the wide projections too (WLX_ROWTILE_NMAX), now that they run four column tiles per workgroup
set -u
TAG=${1:-r4lv3rt}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; export WLX_QUIET=1
prof() { env $1 timeout 600 python scripts/step_profile.py $2 $3 33 2>&1 | sed "s/^==/== [$1]/" | tee -a "$OUT/steps.txt" | head -${4:-12}; }
prof A=1 large-v3 40 12
prof WLX_ROWTILE_NMAX=100000 large-v3 40 12
prof WLX_ROWTILE_NMAX=4000 large-v3 40 4
echo done
