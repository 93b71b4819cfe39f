#!/usr/bin/env python
"""Wall time of a beam-5 decode that ENDS ON AN END-OF-TEXT (every id but the end-of-text suppressed, so all five beams finish in the first step):
prefill + one step + whatever the call waits for after the finish. The benchmark windows end on max_length, where the host loop stops by count and
nothing runs past the end; a transcript's decode ends like this one. usage: [WLX_LIB=...] python scripts/finish_latency.py [model]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from whisperlive_amd.engine import HipWhisperEngine, TokenIds
from whisperlive_amd.specs import get_spec
from whisperlive_amd.synthetic import speech_like_pcm
from whisperlive_amd.weights import random_weights

name = sys.argv[1] if len(sys.argv) > 1 else "small.en"
spec = get_spec(name)
eng = HipWhisperEngine(spec, random_weights(spec, seed=0))
sl = eng.create_slot(1, 5)
T = sl.logmel(speech_like_pcm(30.0, seed=1234))
sl.encode(1, seek=[0], seg=[min(T - 1, 3000)])
ts_begin = spec.vocab - 1501
ids = TokenIds(ts_begin - 106, ts_begin - 107, ts_begin - 1, ts_begin, ts_begin - 2, 220)
everything_but_eot = [i for i in range(spec.vocab) if i != ids.eot]
kw = dict(beam_size=5, max_length=64, suppress_tokens=everything_but_eot, suppress_blank=False)
prompt = [ids.sot, ids.no_timestamps]
for _ in range(5):
    r = sl.generate([prompt], ids, **kw)[0]
assert r.sequences_ids[0] == [], r.sequences_ids
w = []
for _ in range(200):
    t0 = time.perf_counter()
    sl.generate([prompt], ids, **kw)
    w.append(1e3 * (time.perf_counter() - t0))
    time.sleep(0.002)                      # paced: the step behind the finish has drained before the next call
w = np.sort(np.asarray(w))
print(name, "decode that ends on end-of-text in its first step: generate() wall ms p10 / p50 / p90 = %.3f / %.3f / %.3f" % (w[20], w[100], w[180]), flush=True)
sl.close(); eng.close()
os._exit(0)
