"""CPU: the C statement of the VAD segmentation loop (libwlx.so wlx_vad_segments, host only) against the Python one
(whisperlive_amd/vad.py speech_segments_from_probs = faster_whisper.vad.get_speech_timestamps' loop): identical segments on thousands of
seeded probability tracks — uniform noise, speech-like runs, and tracks quantised to 0.05 so that probabilities EQUAL the thresholds —
over the option grid (threshold, min_silence, max_speech incl. inf, speech_pad, min_speech, neg_threshold) and ragged sample counts."""
import numpy as np
import pytest

from whisperlive_amd import vad


def _tracks(rng, n):
    yield rng.random(n).astype(np.float32)
    yield np.clip(np.repeat(rng.random(n // 20 + 1), 20)[:n] * 1.3 - 0.15 + 0.05 * rng.standard_normal(n), 0, 1).astype(np.float32)
    yield (np.round(rng.random(n) * 20) / 20).astype(np.float32)
    yield (np.round(np.repeat(rng.random(n // 7 + 1), 7)[:n] * 20) / 20).astype(np.float32)


def test_native_segmentation_equals_the_python_statement():
    try:
        from whisperlive_amd import _lib
        _lib.load()
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"libwlx.so not built: {e}")
    rng = np.random.default_rng(2026)
    cases = 0
    for trial in range(700):
        n = int(rng.integers(1, 1000)) if trial % 10 else int(rng.integers(1, 4))
        opt = vad.VadOptions(threshold=float(rng.choice([0.5, 0.35, 0.6, 0.05])),
                             neg_threshold=(None if trial % 3 else float(rng.choice([0.1, 0.35, 0.45]))),
                             min_speech_duration_ms=int(rng.choice([0, 250, 1000])),
                             min_silence_duration_ms=int(rng.choice([0, 100, 500, 2000])),
                             max_speech_duration_s=float(rng.choice([float("inf"), 30.0, 3.0, 1.0, 0.1])),
                             speech_pad_ms=int(rng.choice([0, 30, 400, 2000])))
        for p in _tracks(rng, n):
            ns = max(1, n * 512 - int(rng.integers(0, 512)))
            want = vad.speech_segments_from_probs(p, ns, opt)
            got = vad.speech_segments_from_probs_native(p, ns, opt)
            assert got == want, (trial, n, ns, opt, got[:4], want[:4])
            cases += 1
    assert cases == 2800


def test_native_segmentation_edge_tracks():
    try:
        from whisperlive_amd import _lib
        _lib.load()
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"libwlx.so not built: {e}")
    opt = vad.VadOptions(threshold=0.5)
    for p, ns in [(np.zeros(0, np.float32), 0), (np.ones(1, np.float32), 1), (np.ones(940, np.float32), 940 * 512),
                  (np.zeros(300, np.float32), 300 * 512 - 7), (np.tile(np.float32([1, 0]), 400), 800 * 512)]:
        assert vad.speech_segments_from_probs_native(p, ns, opt) == vad.speech_segments_from_probs(p, ns, opt)
