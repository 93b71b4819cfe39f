"""rocprofv3 + hipGraph on this image: WITHOUT the first line (import torch) a traced process that replays the decode-step graph segfaults inside
hipGraphLaunch after ~170 launches (the image's ROCm 7.2 runtime; a memcpy past a page-aligned pool in the runtime / tracer, not in libwlx.so: the
same loop runs for hours untraced, and traced with WLX_NO_GRAPH=1); with torch imported first libwlx.so binds to the HIP runtime torch bundles and
the trace completes. bench.py imports torch. usage: MODE=gen|genonly|timings STEPS=16 CALLS=40 rocprofv3 --kernel-trace ... -- python this"""
import torch  # noqa: F401 (first: libwlx.so then binds to the HIP runtime torch bundles, see scripts/README.md)
import os, sys, time
sys.path.insert(0, os.environ["R"])
import numpy as np
from bench import token_ids, stream_pcm
from whisperlive_amd.engine import HipWhisperEngine, TokenIds
from whisperlive_amd.specs import get_spec
from whisperlive_amd.weights import random_weights
spec = get_spec("small.en")
eng = HipWhisperEngine(spec, random_weights(spec, seed=0))
sl = eng.create_slot(1, 5)
ids = token_ids(spec.vocab); eids = TokenIds(**ids)
pcm = stream_pcm(8.0, 77)
mode = os.environ.get("MODE", "gen")
steps = int(os.environ.get("STEPS", "16"))
for i in range(int(os.environ.get("CALLS", "30"))):
    if mode != "genonly" or i == 0:
        T = sl.logmel(pcm); sl.encode(1, seek=[0], seg=[T - 1])
    sl.generate([[ids["sot"]]], eids, beam_size=5, max_length=1 + steps, suppress_tokens=[ids["eot"]])
    if mode == "timings": sl.timings()
    print("call", i, flush=True)
sl.close(); eng.close()
