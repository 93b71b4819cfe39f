"""Generates tests/golden/jfk_16k.npz from the reference's only audio asset, assets/jfk.flac (the clip of its end-to-end
test, /root/reference/tests/test_server.py:73-118): 24-bit stereo 44.1 kHz FLAC -> mono float32 at 16 kHz through
whisperlive_amd/audio_io.py (FLAC decode checked against the file's own MD5, polyphase resampling). Test infrastructure:
the GPU box has no /root/reference, so the `-m gpu` tests read this fixture. Run in the build container:
    PYTHONPATH=. python tests/golden/make_jfk_fixture.py"""
import hashlib
import os

import numpy as np

from whisperlive_amd import audio_io

SRC = "/root/reference/assets/jfk.flac"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jfk_16k.npz")

if __name__ == "__main__":
    pcm = audio_io.load_audio(SRC, sampling_rate=16000)
    assert pcm.dtype == np.float32 and pcm.ndim == 1
    np.savez_compressed(OUT, pcm=pcm, sampling_rate=np.int32(16000),
                        sha256=np.frombuffer(hashlib.sha256(pcm.tobytes()).digest(), dtype=np.uint8))
    print(OUT, pcm.shape, pcm.shape[0] / 16000.0, "s", os.path.getsize(OUT), "bytes")
