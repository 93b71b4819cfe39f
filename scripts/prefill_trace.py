"""One conditioned window (225-token prompt + N decode steps) per pass, for rocprofv3 --kernel-trace: where does the prompt prefill go?
usage: python scripts/prefill_trace.py [MODEL] [STEPS] [PASSES]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import logmel as olm          # synthetic PCM generator only
from tests import helpers as H
from whisperlive_amd.engine import HipWhisperEngine
from whisperlive_amd.specs import get_spec
from whisperlive_amd.weights import random_weights

name = sys.argv[1] if len(sys.argv) > 1 else "small.en"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
spec = get_spec(name)
eng = HipWhisperEngine(spec, random_weights(spec, seed=0))
ids = H.token_ids_for(spec.vocab)
sl = eng.create_slot(1, 5)
pcm = olm.speech_like_pcm(30.0, seed=1234)
sl.pcm_put(pcm)
prev = np.random.default_rng(5).integers(0, ids.eot, size=223).tolist()
cprompt = [ids.timestamp_begin - 4] + prev + [ids.sot]
kw = dict(beam_size=5, patience=1.0, max_length=len(cprompt) + steps, suppress_tokens=sorted(H.default_suppress(ids) + [ids.eot]))
for i in range(passes):
    T = sl.logmel_resident(0)
    sl.encode(1, seek=[0], seg=[min(T - 1, 3000)])
    sl.generate([cprompt], H.engine_ids(ids), **kw)
    print("pass", i, sl.timings())
sl.close()
eng.close()
