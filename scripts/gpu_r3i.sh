#!/bin/bash
# round 3, call I: multi-wave self-attention — parity (every decode test), headline + conditioned window, decode step by position
set -u
TAG=r3i; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_long_context.py tests/test_gpu_parity.py tests/test_gpu_full_depth.py tests/test_gpu_transcriber.py tests/test_gpu_batched_depth.py -m gpu -q -p no:cacheprovider --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest.log"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-stream > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print({k: d[k] for k in ("value", "ms_per_step")}, d["decode_step"]["graph_replay_ms"], d.get("conditioned_window"))
print([ (k["name"][:40], round(k["avg_us"],2)) for k in d["decode_step"]["kernels"] if "self_attn" in k["name"]])
PY
python - <<'PY' > "$OUT/step_by_position.txt" 2>&1
import sys; sys.path.insert(0, ".")
from oracle import logmel as olm
from whisperlive_amd.engine import HipWhisperEngine
from whisperlive_amd.specs import get_spec
from whisperlive_amd.weights import random_weights
for name in ("small.en", "large-v3"):
    spec = get_spec(name); eng = HipWhisperEngine(spec, random_weights(spec, seed=0)); sl = eng.create_slot(1, 5)
    T = sl.logmel(olm.speech_like_pcm(30.0, seed=1234)); sl.encode(1, seek=[0], seg=[min(T - 1, 3000)])
    print(name, "decode step graph (us) by position:", {t: round(1e3 * sl.debug_time_decode_step(5, t, 30), 1) for t in (8, 33, 63, 100, 200, 300, 447)})
    sl.close(); eng.close()
PY
cat "$OUT/step_by_position.txt" | grep -v amdgpu
du -sh "$OUT"
