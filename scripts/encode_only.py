#!/usr/bin/env python
"""Encoder-only workload for counter passes: Whisper-small shapes, log-mel + encoder (+ cross-K/V) of B 30 s windows (argv: model passes B),
N times. Keeps the dispatch count small (~80 launches per pass) so a --pmc run finishes in seconds."""
import sys

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from oracle import logmel as olm                        # noqa: E402  (synthetic PCM generator only)
from whisperlive_amd.engine import HipWhisperEngine     # noqa: E402
from whisperlive_amd.specs import get_spec              # noqa: E402
from whisperlive_amd.weights import random_weights      # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "small.en"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
spec = get_spec(model)
eng = HipWhisperEngine(spec, random_weights(spec, seed=0), device=0)
slot = eng.create_slot(batch, 5)
pcm = olm.speech_like_pcm(30.0, seed=1234)
for _ in range(n):
    for b in range(batch):
        T = slot.logmel(pcm, item=b)
    slot.encode(batch, seek=[0] * batch, seg=[min(T - 1, 3000)] * batch)
print("model", model, "batch", batch, "encode_ms", slot.timings()["encode_ms"], flush=True)
slot.close()
eng.close()
