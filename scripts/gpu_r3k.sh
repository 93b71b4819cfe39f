#!/bin/bash
# round 3, call K: config 5 by worker lanes; 40-row large-v3 step with a 4-way K split of the MLP output projection
set -u
TAG=r3k; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for L in 1 3 4; do
  timeout 600 python bench.py --config 5 --lanes $L --steps 2 --warmup 1 --no-pmc > "$OUT/bench_config5_lanes$L.json" 2> "$OUT/bench_config5_lanes$L.err"
  python - "$OUT/bench_config5_lanes$L.json" $L <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("config5 lanes", sys.argv[2], "xRT", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), d.get("decode_step", {}).get("graph_replay_ms"))
except Exception as e: print("config5 lanes", sys.argv[2], "FAILED", e)
PY
done
python - <<'PY' 2>&1 | grep -v amdgpu
import os, subprocess, sys
code = r'''
import sys; sys.path.insert(0, ".")
from oracle import logmel as olm
from whisperlive_amd.engine import HipWhisperEngine
from whisperlive_amd.specs import get_spec
from whisperlive_amd.weights import random_weights
import os
spec = get_spec("large-v3"); eng = HipWhisperEngine(spec, random_weights(spec, seed=0)); sl = eng.create_slot(8, 5)
for b in range(8):
    sl.logmel(olm.speech_like_pcm(30.0, seed=1234 + b), b)
sl.encode(8, seek=[0] * 8, seg=[3000] * 8)
print(os.environ.get("WLX_LIB", "libwlx.so"), "large-v3 40-row step (ms):", {t: round(sl.debug_time_decode_step(40, t, 20), 4) for t in (8, 33)}, flush=True)
os._exit(0)
'''
for lib in ("libwlx.so", "libwlx_ks4.so"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WLX_LIB="whisperlive_amd/" + lib))
PY
du -sh "$OUT"
