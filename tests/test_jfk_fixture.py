"""The reference's one audio clip, assets/jfk.flac (its end-to-end test: /root/reference/tests/test_server.py:73-118), as a
committed 16 kHz mono float32 fixture (tests/golden/jfk_16k.npz, made by tests/golden/make_jfk_fixture.py) — REAL speech
through the front of the hot path on the GPU: log-mel (wlx_logmel) and the Silero network (wlx_vad_probs, seeded weights:
none exist offline) against the CPU oracle, the VAD segmentation through either model, and the encoder on it.
The synthetic `speech_like_pcm` of the other tests has neither the dynamic range (the clip has long quiet stretches that
sit on the log-mel clamp `max - 8`) nor the spectral tilt of speech."""
import hashlib
import os

import numpy as np
import pytest

from oracle import logmel as olm
from oracle import silero_vad as sv
from tests import helpers as H

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jfk_16k.npz")
JFK = "/root/reference/assets/jfk.flac"


@pytest.fixture(scope="module")
def jfk():
    z = np.load(FIXTURE)
    pcm = z["pcm"]
    assert pcm.dtype == np.float32 and pcm.shape == (176000,) and int(z["sampling_rate"]) == 16000
    assert hashlib.sha256(pcm.tobytes()).digest() == z["sha256"].tobytes()
    return pcm


def test_fixture_is_the_reference_clip(jfk):
    assert 0.2 < float(np.abs(jfk).max()) <= 1.0 and abs(float(jfk.mean())) < 1e-3
    feats = olm.log_mel_spectrogram(jfk, 80)
    assert feats.shape == (80, 1101) and np.isfinite(feats).all()
    assert float(feats.max() - feats.min()) == pytest.approx(2.0, abs=1e-5)       # the (max - 8) clamp is active: /4 scaling
    if os.path.isfile(JFK):
        from whisperlive_amd import audio_io
        assert np.array_equal(audio_io.load_audio(JFK, sampling_rate=16000), jfk)


@pytest.mark.gpu
def test_jfk_logmel_vad_and_encoder_on_the_gpu(gpu, jfk):
    from whisperlive_amd import vad
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import SPECS
    from whisperlive_amd.weights import random_weights
    from oracle import model as omodel
    spec = SPECS["tiny.en"]
    w = random_weights(spec, seed=7)
    eng = HipWhisperEngine(spec, w)
    slot = eng.create_slot(1, 5)
    try:
        T = slot.logmel(jfk)
        got = slot.features()
        want = olm.log_mel_spectrogram(jfk, spec.n_mels)
        assert got.shape == want.shape == (80, T)
        err = float(np.abs(got - want).max())
        print("jfk log-mel max-abs vs float64 oracle", err, "frames", T)
        assert err <= 2e-4
        slot.encode(1, seek=[0], seg=[T - 1])
        oracle = omodel.WhisperOracle(H.oracle_spec(spec), H.f16_weights(w))
        ref = oracle.encode(olm.pad_or_trim(got[:, : T - 1])[None])[0].numpy()
        st = H.err_stats(slot.encoder_output(0), ref)
        print("jfk encoder (tiny.en shapes, 11 s padded to 30 s)", st)
        assert st["rel_rms"] <= 2e-3, st
    finally:
        slot.close()
        eng.close()
    vw = sv.random_weights(3)
    m = vad.SileroHIPModel(vw, device=0)
    try:
        probs = m(jfk)
        want = sv.speech_probs(vw, jfk)
        assert probs.shape == want.shape == (344,)
        assert float(np.abs(probs - want).max()) <= 2e-5
        thr = float(np.quantile(want, 0.5))
        if np.min(np.abs(want - thr)) > 2e-4 and np.min(np.abs(want - max(thr - 0.15, 0.01))) > 2e-4:
            opt = vad.VadOptions(threshold=thr, min_silence_duration_ms=160, speech_pad_ms=30)
            assert vad.get_speech_timestamps(jfk, opt, model=m) == vad.get_speech_timestamps(jfk, opt, model=lambda x: sv.speech_probs(vw, x))
    finally:
        m.close()
