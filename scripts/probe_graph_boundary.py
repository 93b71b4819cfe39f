import sys, os
sys.path.insert(0, os.getcwd())
import bench
from oracle import logmel as olm
from whisperlive_amd.engine import HipWhisperEngine, TokenIds
from whisperlive_amd.specs import get_spec
from whisperlive_amd.weights import random_weights
spec = get_spec("small.en")
eng = HipWhisperEngine(spec, random_weights(spec, seed=0), device=0)
slot = eng.create_slot(1, 5)
slot.pcm_put(olm.speech_like_pcm(30.0, seed=1234)); T = slot.logmel_resident(); slot.encode(1, seek=[0], seg=[min(T - 1, 3000)])
for _ in range(3):
    print(os.environ.get("WLX_PROBE_PASSES_PER_GRAPH", "1"), "passes/graph:", slot.debug_time_decode_step(rows=5, t=33, iters=40) * 1000, "us per pass")
slot.close(); eng.close()
