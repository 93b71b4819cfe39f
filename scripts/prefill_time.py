"""How long does the prompt prefill of a conditioned window take? (every window after the first carries up to 224 prompt tokens:
whisper_live/transcriber/transcriber_faster_whisper.py:1480-1513). Times wlx_generate with a 225-token prompt and ONE decode step
against the same call with a 1-token prompt; WLX_PREFILL_ROWS selects the chunk size (48 = lean projections, 64 = general kernel).
usage: python scripts/prefill_time.py [model=small.en]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import logmel as olm          # synthetic PCM generator only
from whisperlive_amd.engine import HipWhisperEngine, TokenIds
from whisperlive_amd.specs import get_spec
from whisperlive_amd.weights import random_weights

name = sys.argv[1] if len(sys.argv) > 1 else "small.en"
spec = get_spec(name)
eng = HipWhisperEngine(spec, random_weights(spec, seed=0))
slot = eng.create_slot(1, 5)
T = slot.logmel(olm.speech_like_pcm(30.0, seed=1234))
slot.encode(1, seek=[0], seg=[min(T - 1, 3000)])
tb = spec.vocab - 1501
ids = TokenIds(tb - 106, tb - 107, tb - 1, tb, tb - 2, 220)
prev = np.random.default_rng(5).integers(0, ids.eot, size=223).tolist()
long_p, short_p = [tb - 4] + prev + [ids.sot], [ids.sot]
res = {}
for tag, p in (("1-token prompt", short_p), ("225-token prompt", long_p)):
    kw = dict(beam_size=5, patience=1.0, max_length=len(p) + 1, suppress_tokens=[ids.eot])
    for _ in range(3):
        slot.generate([p], ids, **kw)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); slot.generate([p], ids, **kw); ts.append(time.perf_counter() - t0)
    res[tag] = (1e3 * float(np.median(ts)), slot.timings()["generate_ms"])
    print(f"{name} {tag}: wall {res[tag][0]:.3f} ms, device {res[tag][1]:.3f} ms (prefill + one decode step)")
print(f"{name} prefill of 224 tokens: {res['225-token prompt'][1] - res['1-token prompt'][1]:.3f} ms on the device, "
      f"{res['225-token prompt'][0] - res['1-token prompt'][0]:.3f} ms wall")
slot.close(); eng.close()
