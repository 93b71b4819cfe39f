"""decode-step graph time by decoder position (history length): the self-attention's share grows with t.
usage: [WLX_LIB=whisperlive_amd/libwlx_sa1.so] python scripts/step_by_position.py [model ...]"""
import os
import sys

sys.path.insert(0, ".")
from oracle import logmel as olm          # synthetic PCM generator only
from whisperlive_amd.engine import HipWhisperEngine
from whisperlive_amd.specs import get_spec
from whisperlive_amd.weights import random_weights

for name in (sys.argv[1:] or ["small.en"]):
    spec = get_spec(name)
    eng = HipWhisperEngine(spec, random_weights(spec, seed=0))
    sl = eng.create_slot(1, 5)
    T = sl.logmel(olm.speech_like_pcm(30.0, seed=1234))
    sl.encode(1, seek=[0], seg=[min(T - 1, 3000)])
    print(os.environ.get("WLX_LIB", "libwlx.so"), name, "decode step graph (us) by position:",
          {t: round(1e3 * sl.debug_time_decode_step(5, t, 30), 1) for t in (8, 33, 63, 64, 100, 200, 300, 447)}, flush=True)
    sl.close()
    eng.close()
    os._exit(0) if name == (sys.argv[1:] or ["small.en"])[-1] else None
