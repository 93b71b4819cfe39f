"""Audio file input (whisperlive_amd/audio_io.py): RIFF/WAVE and FLAC readers, down-mix, resampling — and the one audio
clip the reference ships, assets/jfk.flac (24-bit stereo 44.1 kHz, the clip of its end-to-end test, tests/test_server.py:73-118),
decoded here without FFmpeg: the FLAC decoder is pinned by the MD5 of the unencoded audio that the file itself carries.
The jfk tests skip where /root/reference is absent."""
import io
import os
import struct
import wave

import numpy as np
import pytest

from whisperlive_amd import audio_io, vad

JFK = "/root/reference/assets/jfk.flac"
needs_jfk = pytest.mark.skipif(not os.path.isfile(JFK), reason="reference assets not present")


def _wav_bytes(x, sr, width):
    buf = io.BytesIO()
    with wave.open(buf, "wb") as w:
        w.setnchannels(x.shape[1]); w.setsampwidth(width); w.setframerate(sr)
        if width == 2:
            w.writeframes((x * 32767).round().astype("<i2").tobytes())
        elif width == 3:
            v = (x * 8388607).round().astype(np.int32).reshape(-1)
            w.writeframes(b"".join(int(s).to_bytes(3, "little", signed=True) for s in v))
        elif width == 1:
            w.writeframes(((x * 127).round() + 128).astype(np.uint8).tobytes())
    return buf.getvalue()


def test_wav_widths_channels_and_float():
    rng = np.random.default_rng(0)
    x = (rng.random((800, 2)) * 1.6 - 0.8).astype(np.float32)
    for width, tol in ((2, 1e-4), (3, 1e-6), (1, 1.2e-2)):
        y, sr = audio_io.read_wav(_wav_bytes(x, 22050, width))
        assert sr == 22050 and y.shape == x.shape and np.abs(y - x).max() <= tol, width
    raw = x.astype("<f4").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 2, 16000, 16000 * 8, 8, 32)
    y, sr = audio_io.read_wav(hdr + b"LIST" + struct.pack("<I", 4) + b"abcd" + b"data" + struct.pack("<I", len(raw)) + raw)
    assert sr == 16000 and np.array_equal(y, x)                       # float32 payload, an extra chunk skipped
    with pytest.raises(ValueError):
        audio_io.read_wav(b"RIFFxxxxWAVEjunk")
    with pytest.raises(ValueError):
        audio_io.load_audio(b"OggS........")


def test_load_audio_downmix_and_resample_keep_a_tone():
    sr = 44100
    t = np.arange(sr) / sr
    tone = 0.5 * np.sin(2 * np.pi * 440 * t)
    x = np.stack([tone, tone], axis=1).astype(np.float32)
    y = audio_io.load_audio(_wav_bytes(x, sr, 2))
    assert y.dtype == np.float32 and y.shape == (16000,)
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(16000) / 16000)
    assert np.abs(y[200:-200] - ref[200:-200]).max() < 5e-3          # polyphase resampling, away from the edges
    y2 = audio_io.load_audio(io.BytesIO(_wav_bytes(x[:, :1], 16000, 2)))
    assert y2.shape == (sr,) and np.abs(y2 - tone).max() < 1e-4       # already 16 kHz mono: untouched


@needs_jfk
def test_flac_decoder_matches_the_files_own_md5_and_rejects_damage():
    x, sr = audio_io.read_flac(JFK)                                    # raises unless the decode matches STREAMINFO's MD5
    assert (sr, x.shape, x.dtype) == (44100, (485100, 2), np.float32) and 0.5 < np.abs(x).max() < 1.0
    b = bytearray(open(JFK, "rb").read())
    b[200_000] ^= 0x10                                                 # one flipped bit inside a frame
    with pytest.raises(ValueError):
        audio_io.read_flac(bytes(b))


@needs_jfk
def test_jfk_feeds_the_front_end_and_the_vad_gate():
    """the reference's clip through this repo's front half: 11.0 s -> 176 000 samples -> [80, 1101] log-mel (oracle; the
    HIP kernel is held to it on the GPU), and the reference's only VAD assertion (tests/test_vad.py:18-26: speech ->
    detected, silence -> not) on the segmentation logic with the LABELLED energy stand-in (no Silero weights offline)."""
    from oracle import logmel as olm
    pcm = audio_io.load_audio(JFK)
    assert pcm.shape == (176000,) and pcm.dtype == np.float32
    feats = olm.log_mel_spectrogram(pcm, 80)
    assert feats.shape == (80, 1101) and np.isfinite(feats).all() and feats.max() <= 2.0 and feats.min() >= feats.max() - 2.0 - 1e-6
    gate = vad.EnergyGateModel()
    speech = vad.get_speech_timestamps(pcm, vad.VadOptions(threshold=0.5), model=gate)
    assert speech and speech[0]["start"] < 16000 and speech[-1]["end"] > 9 * 16000       # speech from the start to past 9 s
    kept = sum(s["end"] - s["start"] for s in speech)
    assert kept > 0.6 * pcm.size
    assert vad.get_speech_timestamps(np.zeros(176000, np.float32), vad.VadOptions(threshold=0.5), model=gate) == []


@needs_jfk
def test_transcribe_accepts_a_flac_path():
    from tests.fakes import FakeEngine
    from whisperlive_amd.tokenizer import synthetic_tokenizer
    from whisperlive_amd.transcriber import WhisperModelHIP
    eng = FakeEngine()
    tb = eng.spec.vocab - 1501
    eng.default_tokens = [tb, 300, 301, tb + 100]
    m = WhisperModelHIP("fake", engine=eng, hf_tokenizer=synthetic_tokenizer(eng.spec.vocab))
    segs, info = m.transcribe(JFK, language="en")
    assert info.duration == 11.0 and segs and eng.slots[0].calls[0] == ("logmel", 0, 176000)
