"""CPU tests of the HOST side of the device-resident PCM ring (round 6; include/wlx.h wlx_ring_*): the session keeps a mirror of
`frames_np` on its transcriber's GPU in lockstep with the reference's buffer rule (whisper_live/backend/base.py:173-234), hands the
transcriber a `ResidentAudio` for the chunk it took, and `WhisperModelHIP.transcribe` turns the VAD's speech chunks into ring ranges
instead of concatenating and uploading them. The ring here is a numpy stand-in with the library's contract; the kernels' side is
tests/test_gpu_ring.py."""
import json
from unittest.mock import MagicMock

import numpy as np
import pytest

from tests.fakes import FakeEngine, FakeSlot
from whisperlive_amd import vad as wvad
from whisperlive_amd._lib import WlxError
from whisperlive_amd.serve_client import ServeClientHIP
from whisperlive_amd.tokenizer import synthetic_tokenizer
from whisperlive_amd.transcriber import ResidentAudio, WhisperModelHIP

V = 2310


class NumpyRing:
    """wlx_ring_append / wlx_ring_state on the host (absolute positions, 45 s cap / 30 s trim)."""
    device = 0

    def __init__(self):
        self.buf = np.zeros(0, np.float32)
        self.base = 0
        self.closed = False

    def append(self, samples, max_resident=45 * 16000, trim=30 * 16000):
        dropped = 0
        if max_resident > 0 and self.buf.shape[0] > max_resident:
            dropped = min(trim, self.buf.shape[0])
            self.buf = self.buf[dropped:]
            self.base += dropped
        self.buf = np.concatenate([self.buf, np.asarray(samples, np.float32)])
        return dropped, self.base, self.buf.shape[0]

    def state(self):
        return self.base, self.buf.shape[0]

    def read(self, a, b):
        if a < self.base or b > self.base + self.buf.shape[0]:
            raise WlxError("not resident")
        return self.buf[a - self.base: b - self.base]

    def close(self):
        self.closed = True


class RingEngine(FakeEngine):
    def __init__(self):
        super().__init__()
        self.rings = []

    def create_ring(self, capacity_samples=0):
        self.rings.append(NumpyRing())
        return self.rings[-1]

    def create_slot(self, max_batch=1, rows=5):
        s = RingSlot(self, max_batch, rows)
        self.slots.append(s)
        return s


class RingSlot(FakeSlot):
    def logmel_ring(self, ring, ranges, item=0):
        audio = np.concatenate([ring.read(a, b) for a, b in ranges])
        self.calls.append(("logmel_ring", [tuple(map(int, r)) for r in ranges], float(audio.sum())))
        t = (audio.shape[0] + 160) // 160
        self._frames[item] = t
        return t


class ResidentGate(wvad.EnergyGateModel):
    """the energy gate with the device entry point of SileroHIPModel (reads the ring instead of a host array)"""
    device = 0

    def probs_resident(self, ring, start, n):
        x = ring.read(start, start + n)
        return self(np.pad(x, (0, wvad.WINDOW - n % wvad.WINDOW)))


def _speechy(n, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) * 0.2).astype(np.float32)
    x[n // 3: n // 2] = 0.0                       # a silent stretch the gate cuts
    return x


def _client(eng, **kw):
    m = WhisperModelHIP("fake", engine=eng, hf_tokenizer=synthetic_tokenizer(V), vad_model=ResidentGate())
    ws = MagicMock()
    c = ServeClientHIP(ws, client_uid="u", model="small.en", transcriber=m, start_thread=False, **kw)
    return c, m, ws


def test_session_mirror_stays_in_step_with_the_reference_buffer_rule():
    eng = RingEngine()
    c, m, _ = _client(eng, use_vad=False)
    rng = np.random.default_rng(0)
    total = 0
    for i in range(60):                            # 60 x 1.3 s: crosses the 45 s cap twice
        pkt = (rng.standard_normal(20800) * 0.1).astype(np.float32)
        c.add_frames(pkt)
        total += pkt.shape[0]
        ring = eng.rings[0]
        assert c._ring is ring
        base, resident = ring.state()
        assert resident == c.frames_np.shape[0] and base == int(c.frames_offset * 16000) and base + resident == total
        assert np.array_equal(ring.buf, c.frames_np)
    assert c.frames_offset == 60.0                 # two trims of 30 s
    chunk, dur = c.get_audio_chunk_for_processing()
    ra = c._resident(chunk)
    assert isinstance(ra, ResidentAudio) and ra.n == chunk.shape[0] and ra.host is chunk
    assert np.array_equal(eng.rings[0].read(ra.start, ra.start + ra.n), chunk)
    # a chunk that is not the one just taken stays a plain array
    assert c._resident(chunk[:-1]) is not ra and isinstance(c._resident(chunk[:-1]), np.ndarray)
    c.on_transcription_thread_exit()
    assert eng.rings[0].closed and c._ring is False


def test_mirror_failure_switches_it_off_and_the_session_goes_on():
    eng = RingEngine()
    c, m, _ = _client(eng, use_vad=False)
    c.add_frames(np.zeros(16000, np.float32))
    eng.rings[0].append = lambda *a, **k: (_ for _ in ()).throw(WlxError("device lost"))
    c.add_frames(np.ones(16000, np.float32))
    assert c._ring is False and c.frames_np.shape[0] == 32000
    chunk, _ = c.get_audio_chunk_for_processing()
    assert c._resident(chunk) is chunk


def test_no_mirror_without_an_engine_that_has_rings(monkeypatch):
    c, m, _ = _client(FakeEngine(), use_vad=False)
    c.add_frames(np.zeros(16000, np.float32))
    assert c._ring is False
    monkeypatch.setenv("WLX_PCM_RING", "0")
    c2, _, _ = _client(RingEngine(), use_vad=False)
    c2.add_frames(np.zeros(16000, np.float32))
    assert c2._ring is False


@pytest.mark.parametrize("use_vad", [False, True])
def test_transcribe_resident_equals_transcribe_host(use_vad):
    eng = RingEngine()
    tk = synthetic_tokenizer(V)
    m = WhisperModelHIP("fake", engine=eng, hf_tokenizer=tk, vad_model=ResidentGate())
    from whisperlive_amd.tokenizer import Tokenizer
    t = Tokenizer(tk, False)
    eng.default_tokens = [t.timestamp_begin, 300, 301, t.timestamp_begin + 50]
    ring = eng.create_ring()
    lead = np.zeros(12345, np.float32)
    ring.append(lead)
    audio = _speechy(16000 * 9 + 77, 3)
    ring.append(audio)
    kw = dict(language="en", temperature=0.0, vad_filter=use_vad, vad_parameters={"threshold": 0.5, "min_silence_duration_ms": 500})
    want, wi = m.transcribe(audio, **kw)
    slot = eng.slots[-1]
    host_calls = [c for c in slot.calls if c[0] in ("logmel", "logmel_ring")]
    slot.calls.clear()
    got, gi = m.transcribe(ResidentAudio(ring, 12345, audio.shape[0], audio), **kw)
    res_calls = [c for c in slot.calls if c[0] in ("logmel", "logmel_ring")]
    assert [(s.start, s.end, s.tokens) for s in got] == [(s.start, s.end, s.tokens) for s in want]
    assert (gi.duration, gi.duration_after_vad) == (wi.duration, wi.duration_after_vad)
    assert host_calls[0][0] == "logmel" and res_calls[0][0] == "logmel_ring"
    n_host = host_calls[0][2]
    ranges = res_calls[0][1]
    assert sum(b - a for a, b in ranges) == n_host and all(a >= 12345 for a, _ in ranges)
    if use_vad:
        assert n_host < audio.shape[0] and len(ranges) >= 2                   # the silent stretch was cut: two speech ranges
        chunks = wvad.get_speech_timestamps(audio, wvad.VadOptions(threshold=0.5, min_silence_duration_ms=500), model=ResidentGate())
        assert ranges == [(12345 + c["start"], 12345 + c["end"]) for c in chunks]
    else:
        assert ranges == [(12345, 12345 + audio.shape[0])]


def test_transcribe_resident_falls_back_when_the_range_is_gone():
    eng = RingEngine()
    tk = synthetic_tokenizer(V)
    m = WhisperModelHIP("fake", engine=eng, hf_tokenizer=tk, vad_model=ResidentGate())
    ring = eng.create_ring()
    audio = _speechy(16000 * 4, 5)
    ring.append(audio)
    ring.base += 16000                                # as if a trim had dropped the first second
    ring.buf = ring.buf[16000:]
    for use_vad in (False, True):
        segs, info = m.transcribe(ResidentAudio(ring, 0, audio.shape[0], audio), language="en", temperature=0.0, vad_filter=use_vad)
        calls = [c[0] for c in eng.slots[-1].calls if c[0] in ("logmel", "logmel_ring")]
        assert calls[-1] == "logmel" and info.duration == 4.0
