#!/usr/bin/env python
"""Where the HOST side of a chunk goes: cProfile of WhisperModelHIP.transcribe over a scripted engine (tests/fakes.py: computes nothing), the
reference's streaming call (faster_whisper_backend.py:236-244): one 5 s chunk, 16 generated tokens with timestamps, VAD off here (the gate runs on
the GPU in the product). usage: python scripts/host_profile.py [calls]"""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
from fakes import FakeEngine                                   # noqa: E402
from whisperlive_amd.specs import SPECS                        # noqa: E402
from whisperlive_amd.tokenizer import synthetic_tokenizer      # noqa: E402
from whisperlive_amd.transcriber import WhisperModelHIP        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
spec = SPECS["small.en"]
eng = FakeEngine(spec)
tb = spec.vocab - 1501
eng.default_tokens = [tb] + list(range(1000, 1014)) + [tb + 200]          # <|0.00|> 14 text tokens <|4.00|>
m = WhisperModelHIP("fake", engine=eng, hf_tokenizer=synthetic_tokenizer(spec.vocab))
pcm = (0.1 * np.random.default_rng(0).standard_normal(5 * 16000)).astype(np.float32)
kw = dict(language="en", task="transcribe", vad_filter=False, initial_prompt=None)
for _ in range(20):
    segs, info = m.transcribe(pcm, **kw); list(segs)
t0 = time.perf_counter()
for _ in range(n):
    segs, info = m.transcribe(pcm, **kw); list(segs)
print("host side of a call: %.3f ms" % (1e3 * (time.perf_counter() - t0) / n))
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    segs, info = m.transcribe(pcm, **kw); list(segs)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
