"""GPU tests (-m gpu) of the device-resident PCM ring (round 6; include/wlx.h wlx_ring_*): the ring keeps the
reference's buffer rule (whisper_live/backend/base.py:173-234) on the device, the log-mel kernel walks ring ranges and the Silero
front end reads the ring — each against the upload path on the same samples, bit for bit."""
import numpy as np
import pytest

from tests import helpers as H
from whisperlive_amd._lib import WlxError
from whisperlive_amd.synthetic import energy_following_vad_weights, speech_like_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(gpu):
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.weights import random_weights
    e = HipWhisperEngine(H.TINY_EN, random_weights(H.TINY_EN, seed=7))
    yield e
    e.close()


class HostBuffer:
    """ServeClientBase.add_frames (whisperlive_amd/serve_client.py, the reference's base.py:173-198) on a bare numpy buffer"""

    def __init__(self):
        self.buf, self.base = None, 0

    def add(self, pkt):
        dropped = 0
        if self.buf is not None and self.buf.shape[0] > 45 * 16000:
            self.buf, dropped = self.buf[30 * 16000:], 30 * 16000
            self.base += dropped
        self.buf = pkt.copy() if self.buf is None else np.concatenate((self.buf, pkt))
        return dropped


def test_ring_follows_the_reference_buffer_rule_and_holds_the_samples(eng):
    ring = eng.create_ring()
    slot = eng.create_slot(1, 5)
    one = eng.create_slot(1, 5)
    host = HostBuffer()
    rng = np.random.default_rng(1)
    try:
        sizes = [4096] * 40 + [16000 * 7, 123, 16000 * 20, 4096, 16000 * 19, 4096, 16000 * 31, 777, 4096]     # crosses the cap three times, one packet > 30 s
        for i, n in enumerate(sizes):
            pkt = speech_like_pcm(n / 16000.0, seed=100 + i)[:n] if n > 400 else (rng.standard_normal(n) * 0.1).astype(np.float32)
            assert pkt.shape[0] == n
            want_drop = host.add(pkt)
            dropped, base, resident = ring.append(pkt)
            assert (dropped, base, resident) == (want_drop, host.base, host.buf.shape[0]) and ring.state() == (base, resident)
            if i % 9 == 8 or want_drop:
                # the WHOLE resident buffer through the log-mel kernel == the same samples uploaded
                T = slot.logmel_ring(ring, [(base, base + resident)])
                got = slot.features()
                assert one.logmel(host.buf) == T
                assert np.array_equal(got, one.features()), ("packet", i)
        # a range that has been trimmed away is refused, loudly
        with pytest.raises(WlxError, match="resident"):
            slot.logmel_ring(ring, [(0, 16000)])
        with pytest.raises(WlxError, match="ends at"):
            slot.logmel_ring(ring, [(host.base, host.base + host.buf.shape[0] + 1)])
    finally:
        slot.close(); one.close(); ring.close()


@pytest.mark.parametrize("case", ["one", "ragged", "tiny", "many"])
def test_logmel_over_ring_ranges_equals_logmel_of_the_concatenation(eng, case):
    ring = eng.create_ring()
    slot = eng.create_slot(2, 5)
    one = eng.create_slot(1, 5)
    try:
        pcm = speech_like_pcm(40.0, seed=11)
        ring.append(pcm[:300000], max_resident=0)
        ring.append(pcm[300000:], max_resident=0)
        if case == "one":
            ranges = [(1234, 481234)]
        elif case == "ragged":
            ranges = [(0, 41), (41, 17777), (20000, 20200), (100000, 276000), (276001, 300123), (599999, 640000)]
        elif case == "tiny":
            ranges = [(5, 46)]                       # 41 samples: shorter than the reflect padding
        else:
            rng = np.random.default_rng(3)
            cuts = np.sort(rng.choice(np.arange(1, 640000), size=2 * 200, replace=False))
            ranges = [(int(cuts[2 * i]), int(cuts[2 * i + 1])) for i in range(200)]
        cat = np.concatenate([pcm[a:b] for a, b in ranges])
        T = slot.logmel_ring(ring, ranges, item=1)
        assert T == (cat.shape[0] + 160) // 160 == one.logmel(cat)
        got, want = slot.features(1), one.features()
        assert got.shape == want.shape and np.array_equal(got, want)
    finally:
        slot.close(); one.close(); ring.close()


def test_ring_argument_errors(eng):
    ring = eng.create_ring()
    slot = eng.create_slot(1, 5)
    try:
        ring.append(np.zeros(32000, np.float32))
        for bad in ([(100, 100)], [(200, 100)], [(0, 100), (50, 200)], [(-5, 10)]):
            with pytest.raises(WlxError):
                slot.logmel_ring(ring, bad)
        with pytest.raises(WlxError, match="ranges"):
            slot.logmel_ring(ring, [(i * 10, i * 10 + 5) for i in range(257)])
        assert slot.logmel_ring(ring, [(i * 10, i * 10 + 5) for i in range(256)]) == (256 * 5 + 160) // 160
    finally:
        slot.close(); ring.close()


@pytest.mark.parametrize("n", [512 * 40, 512 * 40 + 1, 16000 * 30, 16000 * 30 - 77, 300])
def test_vad_on_the_ring_equals_vad_on_the_upload(eng, n):
    from whisperlive_amd import vad
    vm = vad.SileroHIPModel(energy_following_vad_weights(3), device=0)
    ring = eng.create_ring()
    try:
        pcm = speech_like_pcm(35.0, seed=21)
        ring.append(pcm, max_resident=0)
        start = 4321
        x = pcm[start:start + n]
        want = vm(np.pad(x, (0, vad.WINDOW - n % vad.WINDOW)))
        got = vm.probs_resident(ring, start, n)
        assert got.shape == want.shape == (n // 512 + 1,) and np.array_equal(got, want)
        opt = vad.VadOptions(threshold=0.5)
        assert vad.get_speech_timestamps_resident(ring, start, n, opt, model=vm) == vad.get_speech_timestamps(x, opt, model=vm)
        with pytest.raises(WlxError, match="not resident"):
            vm.probs_resident(ring, 16000 * 35 - 10, 100)
    finally:
        vm.close(); ring.close()


def test_transcribe_resident_audio_end_to_end(eng):
    """WhisperModelHIP.transcribe(ResidentAudio) — VAD on the ring, log-mel over the speech ranges — returns what
    transcribe(host array) returns: same segments, same tokens, same times; and the session feeds it that way."""
    from whisperlive_amd import vad
    from whisperlive_amd.tokenizer import synthetic_tokenizer
    from whisperlive_amd.transcriber import ResidentAudio, WhisperModelHIP
    vm = vad.SileroHIPModel(energy_following_vad_weights(3), device=0)
    m = WhisperModelHIP("rand", engine=eng, hf_tokenizer=synthetic_tokenizer(H.TINY_EN.vocab), vad_model=vm)
    ring = eng.create_ring()
    try:
        pcm = speech_like_pcm(24.0, seed=31)
        t = np.arange(pcm.shape[0]) / 16000.0
        quiet = (t % 12.0) >= 9.0                                            # a 3 s stretch the gate has to cut
        pcm[quiet] = (np.random.default_rng(9).normal(0, 0.003, pcm.shape[0]).astype(np.float32))[quiet]
        ring.append(np.zeros(5000, np.float32))
        ring.append(pcm)
        for use_vad in (True, False):
            kw = dict(language="en", temperature=0.0, max_new_tokens=12, vad_filter=use_vad, vad_parameters={"threshold": 0.5})
            want, wi = m.transcribe(pcm, **kw)
            got, gi = m.transcribe(ResidentAudio(ring, 5000, pcm.shape[0], pcm), **kw)
            assert [(s.start, s.end, s.tokens, s.avg_logprob) for s in got] == [(s.start, s.end, s.tokens, s.avg_logprob) for s in want]
            assert (gi.duration, gi.duration_after_vad) == (wi.duration, wi.duration_after_vad)
            if use_vad:
                assert gi.duration_after_vad < gi.duration
    finally:
        m.close(); vm.close(); ring.close()


def test_ring_readers_against_a_concurrent_writer(eng):
    """The socket thread appends (and trims) while the transcription thread reads: a reader that was launched before a trim must see the samples it
    was launched on (the trim waits for it), a range that is gone must fail loudly, nothing may deadlock. 6 s of wall time: a writer at ~40x real
    time (trims every ~0.75 s), a reader that snapshots a range under the session lock and runs VAD + log-mel on it outside the lock."""
    import threading
    import time
    from whisperlive_amd import vad
    ring = eng.create_ring()
    slot = eng.create_slot(1, 5)
    one = eng.create_slot(1, 5)
    vm = vad.SileroHIPModel(energy_following_vad_weights(3), device=0)
    host = HostBuffer()
    lock = threading.Lock()
    stop = threading.Event()
    errs = []

    def writer():
        try:
            i = 0
            while not stop.is_set():
                pkt = speech_like_pcm(0.256, seed=1000 + i % 50)
                with lock:
                    want = host.add(pkt)
                    dropped, base, resident = ring.append(pkt)
                    assert (dropped, base, resident) == (want, host.base, host.buf.shape[0])
                i += 1
                time.sleep(0.004)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    ok = gone = 0
    t_end = time.time() + 6.0
    rng = np.random.default_rng(5)
    try:
        while time.time() < t_end and not errs:
            with lock:
                if host.buf is None or host.buf.shape[0] < 40000:
                    continue
                n = int(rng.integers(16000, min(host.buf.shape[0], 16000 * 30)))
                off = int(rng.integers(0, host.buf.shape[0] - n + 1))
                start = host.base + off
                ref = host.buf[off: off + n].copy()
            try:
                probs = vm.probs_resident(ring, start, n)
                T = slot.logmel_ring(ring, [(start, start + n // 2), (start + n // 2 + 7, start + n)])
                got = slot.features()
            except WlxError as e:
                assert "resident" in str(e), e
                gone += 1
                continue
            cat = np.concatenate([ref[: n // 2], ref[n // 2 + 7:]])
            assert one.logmel(cat) == T and np.array_equal(got, one.features())
            assert np.array_equal(probs, vm(np.pad(ref, (0, vad.WINDOW - n % vad.WINDOW))))
            ok += 1
    finally:
        stop.set()
        th.join(10)
        vm.close(); slot.close(); one.close(); ring.close()
    assert not errs, errs
    assert not th.is_alive() and ok >= 20, (ok, gone)
    print("ring under a concurrent writer: reads checked", ok, "ranges already trimmed away", gone, "trims", host.base // 480000)
