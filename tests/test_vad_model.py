"""Silero-VAD probability network.

CPU: the numpy restatement (oracle/silero_vad.py) against an independent torch.nn.functional build of the same stack,
the window framing of the reference's contract (whisper_live/vad.py:73-104), the weight-archive loader.
GPU (-m gpu): csrc/vad.hip through the C-ABI (`wlx_vad_*`) against the restatement — tolerance 2e-5 on the
probabilities (fp32 arithmetic in a different summation order; the restatement's own fp32-vs-fp64 gap is 5e-7) —
on ragged lengths, silence, full-scale input, a 45 s buffer, and end to end through get_speech_timestamps.
No Silero weight file exists offline: weights are seeded stand-ins of the exact shapes (parity of the ARCHITECTURE
against the real model is unpinned; see the oracle header)."""
import numpy as np
import pytest

from oracle import logmel as olm
from oracle import silero_vad as sv
from whisperlive_amd import vad

TOL = 2e-5


def test_frame_windows_context_rule():
    x = np.arange(1, 1301, dtype=np.float32)
    f = sv.frame_windows(x)
    assert f.shape == (3, 576)
    assert np.all(f[0, :64] == 0) and np.array_equal(f[0, 64:], x[:512])
    assert np.array_equal(f[1, :64], x[448:512]) and np.array_equal(f[1, 64:], x[512:1024])
    assert np.array_equal(f[2, 64: 64 + 276], x[1024:]) and np.all(f[2, 64 + 276:] == 0)
    assert sv.frame_windows(np.zeros(0, np.float32)).shape == (0, 576)


def test_fourier_basis_is_a_windowed_dft():
    b = sv.fourier_basis()
    assert b.shape == (258, 256)
    x = np.random.default_rng(0).standard_normal(256)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(256) / 256)
    ref = np.fft.rfft(x * win)
    got = b.astype(np.float64) @ x
    assert np.allclose(got[:129], ref.real, atol=1e-4) and np.allclose(got[129:], ref.imag, atol=1e-4)


@pytest.mark.parametrize("n", [1, 512, 513, 5000, 16000 * 3 + 77])
def test_restatement_equals_torch_build(n):
    w = sv.random_weights(11)
    pcm = olm.speech_like_pcm(max(n, 16000) / 16000.0, seed=n)[:n]
    p, q = sv.speech_probs(w, pcm), sv.speech_probs_torch(w, pcm)
    assert p.shape == q.shape == ((n + 511) // 512,)
    assert np.abs(p - q).max() < 1e-6
    assert np.abs(sv.speech_probs(w, pcm, np.float32) - p).max() < 1e-5
    assert sv.speech_probs(w, np.zeros(0, np.float32)).shape == (0,)


def test_probabilities_respond_to_signal_and_state():
    """The stand-in weights must exercise the network: probabilities spread over (0,1), depend on the audio and on
    the recurrent state (a window's probability changes with what preceded it)."""
    w = sv.random_weights(3)
    pcm = olm.speech_like_pcm(10.0, seed=5)
    p = sv.speech_probs(w, pcm)
    assert p.min() > 0 and p.max() < 1 and p.std() > 0.05
    assert np.abs(sv.speech_probs(w, pcm[512 * 10:])[:20] - p[10:30]).max() > 1e-3


def test_weight_archive_loader(tmp_path):
    w = sv.random_weights(1)
    exported = dict(w)
    exported["stft_basis"] = w["stft_basis"].reshape(258, 1, 256)        # torch keeps the singleton conv axes
    exported["out_w"] = w["out_w"].reshape(1, 128, 1)
    np.savez(tmp_path / "silero.npz", **exported)
    got = vad.load_silero_npz(str(tmp_path / "silero.npz"))
    assert set(got) == set(vad.SILERO_SHAPES)
    for k, a in got.items():
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.shape == vad.SILERO_SHAPES[k]
        assert np.array_equal(a.reshape(-1), w[k].reshape(-1))
    bad = dict(w); del bad["lstm_w_hh"]
    with pytest.raises(ValueError, match="lstm_w_hh"):
        vad.check_silero_weights(bad)
    bad = dict(w); bad["enc1_w"] = np.zeros((64, 128, 5), np.float32)
    with pytest.raises(ValueError, match="enc1_w"):
        vad.check_silero_weights(bad)


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hip_vad(gpu):
    w = sv.random_weights(3)
    m = vad.SileroHIPModel(w, device=0)
    yield m, w
    m.close()


def _padded(pcm):
    return np.pad(pcm.astype(np.float32), (0, 512 - pcm.shape[0] % 512))     # get_speech_timestamps' padding rule


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 511, 512, 513, 1024, 2047, 5000, 16000, 16000 * 7 + 333])
def test_hip_probs_match_restatement_ragged(hip_vad, n):
    m, w = hip_vad
    pcm = olm.speech_like_pcm(max(n, 16000) / 16000.0, seed=100 + n)[:n]
    got = m(pcm)                                       # unpadded: the kernel zero-fills the last window itself
    want = sv.speech_probs(w, pcm)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < TOL, np.abs(got - want).max()
    assert np.array_equal(m(_padded(pcm))[: want.shape[0]], got)          # padding on the host changes nothing


@pytest.mark.gpu
def test_hip_probs_edge_inputs(hip_vad):
    m, w = hip_vad
    assert m(np.zeros(0, np.float32)).shape == (0,)
    for pcm in (np.zeros(16000, np.float32), np.ones(8000, np.float32), -np.ones(8000, np.float32),
                np.random.default_rng(0).uniform(-1, 1, 24000).astype(np.float32)):
        got, want = m(pcm), sv.speech_probs(w, pcm)
        assert np.all(np.isfinite(got)) and np.abs(got - want).max() < TOL


@pytest.mark.gpu
def test_hip_probs_full_buffer_deterministic_and_timed(hip_vad):
    """45 s (the session buffer cap, whisper_live/backend/base.py:173-203) and 30 s: parity over ~1400 recurrent steps,
    bit-identical on repetition, and the device time of a 30 s chunk recorded."""
    m, w = hip_vad
    pcm = olm.speech_like_pcm(45.0, seed=9)
    got = m(pcm)
    assert got.shape == (1407,)
    assert np.abs(got - sv.speech_probs(w, pcm)).max() < TOL
    assert np.array_equal(m(pcm), got)
    m(pcm[: 30 * 16000])
    assert 0 < m.last_device_ms < 5.0, m.last_device_ms
    longer = olm.speech_like_pcm(100.0, seed=10)             # beyond the initial 64 s reservation: buffers grow
    assert np.abs(m(longer) - sv.speech_probs(w, longer)).max() < TOL


@pytest.mark.gpu
def test_speech_timestamps_identical_through_either_model(hip_vad):
    m, w = hip_vad
    pcm = np.concatenate([olm.speech_like_pcm(4.0, seed=1), np.zeros(3 * 16000, np.float32), olm.speech_like_pcm(5.0, seed=2)])
    probs = sv.speech_probs(w, _padded(pcm))
    for thr in (float(np.quantile(probs, 0.4)), float(np.quantile(probs, 0.7))):
        if np.min(np.abs(probs - thr)) < 10 * TOL or np.min(np.abs(probs - max(thr - 0.15, 0.01))) < 10 * TOL:
            continue                                        # a probability sits on the decision boundary
        opt = vad.VadOptions(threshold=thr, min_silence_duration_ms=200, speech_pad_ms=100)
        a = vad.get_speech_timestamps(pcm, opt, model=m)
        b = vad.get_speech_timestamps(pcm, opt, model=lambda x: sv.speech_probs(w, x))
        assert a == b
