#!/bin/bash
# round 4, call B: row-tile policy experiments — which projections to cut into row tiles (N limit), tile height, K-split off
set -u
TAG=r4b; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
run() { # env... -- model rows
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python scripts/step_profile.py "$@" 2>&1 | tee -a "$OUT/steps.txt" | head -${LINES_OUT:-3}
}
LINES_OUT=14
run WLX_FC2_KS=0 -- small.en 60
run WLX_ROWTILE_NMAX=1000 -- small.en 60
run WLX_ROWTILE_CHUNK=32 -- small.en 60
LINES_OUT=3
run WLX_FC2_KS=0 -- small.en 40
run WLX_ROWTILE_NMAX=1000 -- small.en 40
run WLX_ROWTILE_CHUNK=32 -- small.en 40
run WLX_ROWTILE_NMAX=1000 -- small.en 20
run WLX_ROWTILE_NMAX=1000 WLX_FC2_KS=0 -- small.en 60
run WLX_ROWTILE_NMAX=3100 -- small.en 60
LINES_OUT=14
run WLX_ROWTILE_NMAX=1300 -- large-v3 40
run WLX_ROWTILE_NMAX=1300 WLX_FC2_KS=0 -- large-v3 40
run WLX_ROWTILE_CHUNK=32 -- large-v3 40
