"""CPU tests of the callers on either side of the hot path: the per-client session state machine (contract of the
reference's tests/test_base_backend.py: buffer cap/trim, chunk slicing, clip rule, commit logic, first-frame wait),
the HIP backend adaptor, and the batch worker (contract of tests/test_batch_inference.py: single -> transcribe,
multi -> encode + generate, error propagation, worker survival, max_batch_size)."""
import json
import threading
import time
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import pytest

from tests.fakes import FakeEngine
from whisperlive_amd import metrics
from whisperlive_amd.batching import BatchInferenceWorker, BatchRequest
from whisperlive_amd.engine import GenerationResult
from whisperlive_amd.serve_client import ServeClientBase, ServeClientHIP
from whisperlive_amd.tokenizer import Tokenizer, synthetic_tokenizer
from whisperlive_amd.transcriber import WhisperModelHIP
from whisperlive_amd.vad import EnergyGateModel

V = 2310


class Client(ServeClientBase):
    def __init__(self, **kw):
        super().__init__("uid", MagicMock(), **kw)
        self.language = "en"
        self.script = []

    def transcribe_audio(self, x):
        return self.script.pop(0) if self.script else None

    def handle_transcription_output(self, result, duration):
        seg = self.update_segments(result, duration)
        self.send_transcription_to_client(self.prepare_segments(seg))


def seg(start, end, text, nsp=0.0):
    return SimpleNamespace(start=start, end=end, text=text, no_speech_prob=nsp)


def test_buffer_cap_trim_and_chunk_slicing():
    c = Client()
    c.add_frames(np.ones(16000, np.float32))
    c.add_frames(np.ones(8000, np.float32) * 2)
    assert c.frames_np.shape[0] == 24000 and c.frames_np[16000] == 2
    c.frames_np = np.zeros(46 * 16000, np.float32)
    c.add_frames(np.ones(16000, np.float32))
    assert c.frames_offset == 30.0 and c.frames_np.shape[0] == 17 * 16000 and c.timestamp_offset == 30.0
    c.timestamp_offset = 35.0
    chunk, dur = c.get_audio_chunk_for_processing()
    assert chunk.shape[0] == 12 * 16000 and dur == 12.0
    # clip rule: > 25 s unconsumed -> keep the last 5 s
    c2 = Client(clip_audio=True)
    c2.frames_np = np.zeros(28 * 16000, np.float32)
    c2.clip_audio_if_no_valid_segment()
    assert c2.timestamp_offset == 23.0


def test_add_frames_thread_safety():
    c = Client()
    ths = [threading.Thread(target=lambda: [c.add_frames(np.ones(160, np.float32)) for _ in range(50)]) for _ in range(8)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert c.frames_np.shape[0] == 8 * 50 * 160


def test_update_segments_commits_all_but_last_and_advances_offset():
    c = Client()
    last = c.update_segments([seg(0.0, 2.0, " one"), seg(2.0, 4.0, " two"), seg(4.0, 5.5, " thr")], 6.0)
    assert [s["text"] for s in c.transcript] == [" one", " two"] and all(s["completed"] for s in c.transcript)
    assert c.timestamp_offset == 4.0 and last == {"start": "4.000", "end": "5.500", "text": " thr", "completed": False}
    # high no-speech tail: nothing is committed, nothing is sent as partial
    c2 = Client()
    assert c2.update_segments([seg(0, 1, " a"), seg(1, 2, " b", nsp=0.9)], 3.0) is None and c2.transcript == []
    # segment end is clamped to the chunk duration
    c3 = Client()
    c3.update_segments([seg(0.0, 9.0, " long"), seg(9.0, 9.5, " t")], 5.0)
    assert c3.transcript[0]["end"] == "5.000" and c3.timestamp_offset == 5.0


def test_repeated_output_is_committed_after_threshold(monkeypatch):
    monkeypatch.setattr(time, "sleep", lambda s: None)
    c = Client(same_output_threshold=3)
    for i in range(5):
        out = c.update_segments([seg(0.0, 1.5 + 0.1 * i, " same")], 3.0)
    assert c.transcript and c.transcript[-1]["text"] == " same" and c.transcript[-1]["completed"]
    assert c.transcript[-1]["end"] == "1.600"              # end time captured when the repetition STARTED
    assert c.timestamp_offset == pytest.approx(1.6) and out is None and c.same_output_count == 0
    assert c.prepare_segments({"x": 1})[-1] == {"x": 1}
    c.send_last_n_segments = 1
    c.transcript = [{"a": 1}, {"b": 2}]
    assert c.prepare_segments() == [{"b": 2}]


def test_speech_to_text_loop_measures_latency_and_handles_errors(monkeypatch):
    metrics.snapshot(reset=True)
    c = Client()
    c.script = [[seg(0.0, 1.0, " a"), seg(1.0, 1.5, " b")], None]
    c.add_frames(np.zeros(2 * 16000, np.float32))
    th = threading.Thread(target=c.speech_to_text, daemon=True)
    th.start()
    time.sleep(0.6)
    c.cleanup(); th.join(timeout=2)
    snap = metrics.snapshot()
    assert snap["chunks"] == 1 and snap["audio_s"] == 2.0 and snap["xrt"] > 0
    sent = json.loads(c.websocket.send.call_args_list[0][0][0])
    assert sent["uid"] == "uid" and sent["segments"][0]["completed"] and sent["segments"][-1]["text"] == " b"
    assert c.timestamp_offset >= 1.0                       # advanced by the commit (then by the silent chunk)
    # a client with no frames idles cheaply and exits on cleanup
    idle = Client()
    t2 = threading.Thread(target=idle.speech_to_text, daemon=True); t2.start(); time.sleep(0.15); idle.cleanup(); t2.join(timeout=1)
    assert not t2.is_alive()


def _model(engine=None):
    return WhisperModelHIP("fake", engine=engine or FakeEngine(), hf_tokenizer=synthetic_tokenizer(V), max_batch=8,
                           vad_model=EnergyGateModel())      # explicit, labelled stand-in: the host logic is what is tested


def test_serve_client_hip_protocol_and_model_failure():
    ws = MagicMock()
    m = _model()
    tk = Tokenizer(m.hf_tokenizer, False)
    m.engine.default_tokens = [tk.timestamp_begin] + tk.encode(" hi there") + [tk.timestamp_begin + 40]
    c = ServeClientHIP(ws, client_uid="u1", model="small.en", transcriber=m, start_thread=False, use_vad=False)
    ready = json.loads(ws.send.call_args_list[0][0][0])
    assert ready == {"uid": "u1", "message": "SERVER_READY", "backend": "faster_whisper"} and c.language == "en"
    out = c.transcribe_audio(np.zeros(16000, np.float32) + 0.01)
    assert out[0].text == " hi there"
    c.handle_transcription_output(out, 1.0)
    msg = json.loads(ws.send.call_args_list[-1][0][0])
    assert msg["segments"][0]["text"] == " hi there" and msg["segments"][0]["completed"] is False
    # model load failure -> ERROR status + close (faster_whisper_backend.py:108-116)
    ws2 = MagicMock()
    ServeClientHIP.MODELS.clear()
    ServeClientHIP(ws2, client_uid="u2", model="/no/such/dir", start_thread=False)
    err = json.loads(ws2.send.call_args_list[0][0][0])
    assert err["status"] == "ERROR" and "Failed to load model" in err["message"] and ws2.close.called
    # language detection message
    ws3 = MagicMock()
    ml = WhisperModelHIP("fake", engine=FakeEngine(), hf_tokenizer=synthetic_tokenizer(V), multilingual=True)
    c3 = ServeClientHIP(ws3, client_uid="u3", model="small", transcriber=ml, start_thread=False, use_vad=False)
    assert c3.language is None
    c3.transcribe_audio(np.zeros(16000, np.float32) + 0.01)
    assert c3.language == "en" and json.loads(ws3.send.call_args_list[-1][0][0])["language"] == "en"


# ------------------------------------------------------------------------------------------------ batch worker
def _mock_transcriber():
    """Same shape as the reference's fixture (tests/test_batch_inference.py:52-78)."""
    t = MagicMock()
    t.feature_extractor.sampling_rate = 16000
    t.feature_extractor.side_effect = lambda a: np.zeros((80, (len(a) + 160) // 160), np.float32)
    del t.encode_audio_batch                                  # force the generic duck-typed path
    t.encode.side_effect = lambda f: np.zeros((f.shape[0], 1500, 512), np.float32)
    t.model.is_multilingual = False
    t.hf_tokenizer = synthetic_tokenizer(51864)
    t.max_length, t.frames_per_second = 448, 100
    t.get_prompt.return_value = [50257]
    t.model.generate.side_effect = lambda enc, prompts, **kw: [
        SimpleNamespace(sequences_ids=[[50363, 1234, 50463]], scores=[-0.1], no_speech_prob=0.01) for _ in prompts]
    t._split_segments_by_timestamps.side_effect = lambda **kw: ([dict(seek=0, start=0.0, end=2.0, tokens=kw["tokens"])], 0, True)
    return t


def test_batch_single_request_uses_transcribe():
    t = _mock_transcriber()
    t.transcribe.return_value = ([SimpleNamespace(text="x")], SimpleNamespace(language="en"))
    w = BatchInferenceWorker(t, max_batch_size=4, batch_window_ms=10); w.start()
    r = BatchRequest(audio=np.zeros(16000, np.float32), use_vad=False); w.submit(r)
    assert r.future.wait(2) and r.error is None and r.result[0].text == "x"
    t.transcribe.assert_called_once(); t.encode.assert_not_called(); w.stop()


def test_batch_multi_uses_one_encode_and_one_generate():
    t = _mock_transcriber()
    w = BatchInferenceWorker(t, max_batch_size=8, batch_window_ms=200)
    reqs = [BatchRequest(audio=np.zeros(16000 * (i + 1), np.float32) + 0.01, use_vad=False) for i in range(3)]
    w._process_batch(reqs)
    assert all(r.future.is_set() and r.error is None for r in reqs)
    t.transcribe.assert_not_called(); assert t.encode.call_count == 1 and t.model.generate.call_count == 1
    assert t.encode.call_args[0][0].shape == (3, 80, 3000)
    kw = t.model.generate.call_args[1]
    assert kw["beam_size"] == 5 and kw["sampling_temperature"] == 0.0 and "sampling_topk" not in kw
    assert reqs[0].result[0].tokens == [50363, 1234, 50463] and reqs[1].info.language == "en"


def test_batch_error_propagation_and_worker_survival():
    t = _mock_transcriber()
    t.encode.side_effect = RuntimeError("boom")
    w = BatchInferenceWorker(t, max_batch_size=2, batch_window_ms=100); w.start()
    rs = [BatchRequest(audio=np.zeros(16000, np.float32), use_vad=False) for _ in range(2)]
    [w.submit(r) for r in rs]
    assert all(r.future.wait(3) for r in rs) and all(isinstance(r.error, RuntimeError) for r in rs)
    t.transcribe.return_value = ([], SimpleNamespace(language="en"))
    r = BatchRequest(audio=np.zeros(16000, np.float32), use_vad=False); w.submit(r)
    assert r.future.wait(3) and r.error is None              # the worker thread is still alive
    w.stop(); assert not w._thread.is_alive()


def test_batch_respects_max_batch_size():
    t = _mock_transcriber()
    sizes = []
    w = BatchInferenceWorker(t, max_batch_size=2, batch_window_ms=300)
    orig = w._process_batch
    w._process_batch = lambda b: (sizes.append(len(b)), orig(b))[1]
    rs = [BatchRequest(audio=np.zeros(16000, np.float32) + 0.01, use_vad=False) for _ in range(5)]
    [w.submit(r) for r in rs]
    w.start()
    assert all(r.future.wait(5) for r in rs)
    w.stop(); assert max(sizes) <= 2 and sum(sizes) == 5


def test_batch_fallback_retries_only_failed_items_on_the_hip_transcriber():
    eng = FakeEngine()
    m = _model(eng)
    tk = Tokenizer(m.hf_tokenizer, False)
    tb = tk.timestamp_begin
    good = [tb] + tk.encode(" fine") + [tb + 50]
    bad = [tb] + tk.encode(" la" * 150) + [tb + 50]
    eng.generate_script = [
        lambda p, i, k: [GenerationResult([good], [-0.1], 0.0), GenerationResult([bad], [-0.1], 0.0), GenerationResult([good], [-0.1], 0.0)],
        lambda p, i, k: [GenerationResult([good], [-0.3], 0.0)],
    ]
    w = BatchInferenceWorker(m, max_batch_size=8)
    reqs = [BatchRequest(audio=np.zeros(16000 * 2, np.float32) + 0.01, use_vad=False) for _ in range(3)]
    w._process_batch(reqs)
    assert all(r.error is None for r in reqs)
    calls = eng.slots[0].calls
    assert [c[0] for c in calls] == ["logmel", "logmel", "logmel", "encode", "generate", "generate"]   # no re-encode
    assert calls[3] == ("encode", 3, [0, 0, 0], [201, 201, 201])                  # trailing pad frame kept (quirk)
    assert calls[5][2]["enc_items"] == [1] and calls[5][2]["sampling_temperature"] == 0.0 and calls[5][2]["beam_size"] == 1
    assert reqs[1].result[0].temperature == 0.2 and reqs[0].result[0].temperature == 0.0
    # empty-after-VAD item is answered without touching the engine
    silent = BatchRequest(audio=np.zeros(16000, np.float32), use_vad=True, vad_parameters={"threshold": 0.5})
    w._process_batch([silent, BatchRequest(audio=np.zeros(16000, np.float32) + 0.01, use_vad=False)])
    assert silent.future.is_set() and silent.result == []


def test_hip_backend_goes_through_the_batch_worker():
    m = _model()
    tk = Tokenizer(m.hf_tokenizer, False)
    m.engine.default_tokens = [tk.timestamp_begin] + tk.encode(" batched") + [tk.timestamp_begin + 40]
    # wait_when_idle=True: the reference's collection (always wait out the window), so the three requests form ONE batch
    w = BatchInferenceWorker(m, max_batch_size=4, batch_window_ms=150, wait_when_idle=True); w.start()
    ServeClientHIP.BATCH_WORKER = w
    try:
        cs = [ServeClientHIP(MagicMock(), client_uid=f"c{i}", model="small.en", transcriber=m, start_thread=False, use_vad=False)
              for i in range(3)]
        outs = [None] * 3
        ths = [threading.Thread(target=lambda i=i: outs.__setitem__(i, cs[i].transcribe_audio(np.zeros(16000, np.float32) + 0.01)))
               for i in range(3)]
        [t.start() for t in ths]; [t.join(5) for t in ths]
        assert all(o and o[0].text == "batched" for o in outs)
    finally:
        ServeClientHIP.BATCH_WORKER = None
        w.stop()


def test_engine_slots_are_pooled_across_client_threads():
    """One slot per CONCURRENT client thread, not per connection ever made: a slot whose owner thread has ended (or that
    was released) is taken over by the next thread that needs one."""
    eng = FakeEngine()
    m = _model(eng)
    seen = []

    def session(hold=None, release=False):
        s = m._slot()
        seen.append(s)
        assert m._slot() is s                       # stable within a thread
        if hold is not None:
            hold.wait(5)
        if release:
            m.release_slot()

    for _ in range(3):                              # three consecutive connections: one slot, reused
        t = threading.Thread(target=session)
        t.start(); t.join()
    assert len(eng.slots) == 1 and seen[0] is seen[1] is seen[2]
    gate = threading.Event()                        # two concurrent connections: a second slot, no more
    ts = [threading.Thread(target=session, args=(gate,)) for _ in range(2)]
    [t.start() for t in ts]
    deadline = time.time() + 5
    while len(seen) < 5 and time.time() < deadline:
        time.sleep(0.01)
    assert len(eng.slots) == 2 and seen[3] is not seen[4]
    gate.set(); [t.join() for t in ts]
    # an explicit release makes the slot available while its thread lives on
    done, keep = threading.Event(), threading.Event()

    def releasing():
        session(release=True); done.set(); keep.wait(5)
    t1 = threading.Thread(target=releasing); t1.start(); done.wait(5)
    t2 = threading.Thread(target=session); t2.start(); t2.join()
    keep.set(); t1.join()
    assert len(eng.slots) == 2
    # a closed slot is never handed out again
    for s in eng.slots:
        s.close()
    t = threading.Thread(target=session); t.start(); t.join()
    assert len(eng.slots) == 3 and seen[-1].sid >= 0


def test_wide_batches_decode_in_groups_over_the_same_encoder_output():
    """One decode step covers every row the slot holds (round 5: max_batch x 5 rows, up to 320; it was 64 rows = 12 clips at beam 5, and
    48 until round 4, when a 12-item beam-5 batch was split 9 + 3): 12 and 24 items x 5 beams are ONE decode each; a call past the step
    limit is decoded as groups (here 9 + 3 items at a 48-row limit) over the SAME encoder output (item maps), one result per prompt, in
    order; more rows per item than the slot was built for is refused with the limit spelled out."""
    eng = FakeEngine()
    tb = eng.spec.vocab - 1501
    seen = []

    def script(prompts, ids, kw):
        seen.append((len(prompts), kw.get("enc_items")))
        return [GenerationResult([[tb, 300 + i, tb + 50]], [-0.1], 0.01) for i in range(len(prompts))]

    eng.generate_script = [script]
    m = WhisperModelHIP("fake", engine=eng, hf_tokenizer=synthetic_tokenizer(V), max_batch=12, vad_model=EnergyGateModel())
    enc = m.encode(np.zeros((12, 80, 3000), np.float32))
    tk = Tokenizer(m.hf_tokenizer, False)
    res = m.model.generate(enc, [[tk.sot]] * 12, beam_size=5)
    assert seen == [(12, None)] and len(res) == 12    # fits one decode: no item map needed
    assert eng.slots[-1].max_batch == 12 and eng.slots[-1].rows == 5
    seen.clear()
    eng.generate_script = [script, script]
    enc = m.encode(np.zeros((12, 80, 3000), np.float32))
    m.model.MAX_DEC_ROWS = 48                          # (the grouping itself, at the pre-round-4 step limit)
    res = m.model.generate(enc, [[tk.sot]] * 12, beam_size=5)
    m.model.MAX_DEC_ROWS = 320
    assert seen == [(9, list(range(9))), (3, [9, 10, 11])] and len(res) == 12
    with pytest.raises(ValueError, match="5 decoder rows per audio item"):
        m.model.generate(enc, [[tk.sot]] * 12, beam_size=7)          # more rows per item than the slot holds: refused with the limit spelled out
    assert [r.sequences_ids[0][1] for r in res] == [300 + i for i in range(9)] + [300 + i for i in range(3)]
    seen.clear()
    eng.generate_script = [script]
    enc8 = m.encode(np.zeros((8, 80, 3000), np.float32))
    m.model.generate(enc8, [[tk.sot]] * 8, beam_size=5)
    assert seen == [(8, None)]
    # a 24-clip transcriber (the reference's worker takes any max_batch_size, batch_inference.py:113-121): slots of 24 items x 5 rows,
    # the whole batch in ONE decode
    seen.clear()
    eng.generate_script = [script]
    m24 = WhisperModelHIP("fake", engine=eng, hf_tokenizer=synthetic_tokenizer(V), max_batch=24, vad_model=EnergyGateModel())
    enc24 = m24.encode(np.zeros((24, 80, 3000), np.float32))
    res = m24.model.generate(enc24, [[tk.sot]] * 24, beam_size=5)
    assert seen == [(24, None)] and len(res) == 24
    assert eng.slots[-1].max_batch == 24 and eng.slots[-1].rows == 5


def test_slot_widens_for_a_larger_beam_and_is_then_kept():
    """`_slot(rows)`: five rows per item by default; a transcribe call with beam_size 8 replaces the thread's slot by an 8-row one (closed
    slots leave the pool), later calls with the default beam keep it; more than the engine's 16 rows per item is refused."""
    eng = FakeEngine()
    m = WhisperModelHIP("fake", engine=eng, hf_tokenizer=synthetic_tokenizer(V), vad_model=EnergyGateModel())
    s5 = m._slot()
    assert (s5.max_batch, s5.rows) == (1, 5) and m._slot() is s5
    s8 = m._slot(rows=8)
    assert s8 is not s5 and s8.rows == 8 and s5.sid < 0 and m._slots == [s8]
    assert m._slot() is s8 and m._slot(rows=6) is s8
    with pytest.raises(ValueError, match="16 rows"):
        m._slot(rows=17)
    # a refused request leaves the thread's slot alone (it used to be closed before the check)
    assert s8.sid >= 0 and m._slots == [s8] and m._slot() is s8
    m.max_batch = 64
    with pytest.raises(ValueError, match="320"):
        m._slot(rows=9)
    assert s8.sid >= 0 and m._slot() is s8 and m._slot(rows=6) is s8


def test_vad_unavailable_from_a_factory_built_transcriber_downgrades_once(monkeypatch):
    """a model_factory-built transcriber without Silero weights raised VadUnavailable on EVERY chunk (the server's
    availability check only runs for transcribers it builds itself): the session must downgrade to use_vad=False once, warn
    the client, and keep producing text."""
    import json
    from unittest.mock import MagicMock
    from whisperlive_amd import vad
    from whisperlive_amd.serve_client import ServeClientHIP
    from whisperlive_amd.types import Segment, TranscriptionInfo

    calls = []

    class Tr:
        def transcribe(self, audio, **kw):
            calls.append(kw.get("vad_filter"))
            if kw.get("vad_filter"):
                raise vad.VadUnavailable("no weights")
            seg = Segment(id=1, seek=0, start=0.0, end=1.0, text=" hi", tokens=[1], avg_logprob=-0.1, compression_ratio=1.0,
                          no_speech_prob=0.0, words=None, temperature=0.0)
            return [seg], TranscriptionInfo(language="en", language_probability=1.0, duration=1.0, duration_after_vad=1.0,
                                            transcription_options=None, vad_options=None, all_language_probs=None)

    ws = MagicMock()
    c = ServeClientHIP.__new__(ServeClientHIP)
    c.websocket, c.client_uid, c.transcriber, c.use_vad, c.serialize = ws, "u1", Tr(), True, False
    c.language, c.task, c.initial_prompt, c.vad_parameters, c.hotwords, c.word_timestamps, c.device_index = "en", "transcribe", None, {"threshold": 0.5}, None, False, 0
    saved = ServeClientHIP.BATCH_WORKER, dict(ServeClientHIP.BATCH_WORKERS)
    ServeClientHIP.BATCH_WORKER = None
    ServeClientHIP.BATCH_WORKERS.clear()
    try:
        out = c.transcribe_audio(np.zeros(16000, np.float32))
        assert [s.text for s in out] == [" hi"] and c.use_vad is False and calls == [True, False]
        out = c.transcribe_audio(np.zeros(16000, np.float32))
        assert calls == [True, False, False]                                   # no second failure, no second warning
        msgs = [json.loads(a[0][0]) for a in ws.send.call_args_list]
        assert [m["status"] for m in msgs if "status" in m] == ["WARNING"]
    finally:
        ServeClientHIP.BATCH_WORKER = saved[0]
        ServeClientHIP.BATCH_WORKERS.update(saved[1])


def test_batch_worker_lanes_overlap_consecutive_batches():
    """lanes=2: a request that arrives while a batch is on the GPU is collected and run by the second lane instead of waiting
    for the first batch to finish; collection itself stays serial (batches form as with one lane)."""
    import threading
    import time
    from whisperlive_amd.batching import BatchInferenceWorker, BatchRequest

    class SlowTr:
        max_batch = 4

        def __init__(self):
            self.active, self.peak, self.lock = 0, 0, threading.Lock()

        def transcribe(self, audio, **kw):
            with self.lock:
                self.active += 1
                self.peak = max(self.peak, self.active)
            time.sleep(0.15)
            with self.lock:
                self.active -= 1
            return [], None

    def run(lanes):
        tr = SlowTr()
        w = BatchInferenceWorker(tr, max_batch_size=4, batch_window_ms=5, lanes=lanes)
        w.start()
        try:
            a, b = BatchRequest(audio=np.zeros(1600, np.float32), use_vad=False), BatchRequest(audio=np.zeros(1600, np.float32), use_vad=False)
            t0 = time.monotonic()
            w.submit(a)
            time.sleep(0.04)                       # after a's 5 ms window closed: b is a second batch
            w.submit(b)
            assert a.future.wait(2) and b.future.wait(2)
            return time.monotonic() - t0, tr.peak
        finally:
            w.stop()
    t1, peak1 = run(1)
    t2, peak2 = run(2)
    assert peak1 == 1 and peak2 == 2
    assert t1 > 0.29 and t2 < 0.26, (t1, t2)


def test_batch_worker_starts_at_once_when_idle_and_batches_while_busy():
    """The collection window only pays while the GPU is busy (the default, wait_when_idle=False): an idle worker takes a lone
    request at once; requests that arrive while a lane is busy are collected over the window and batched. Deterministic: the
    busy lane is HELD by an event, the window is far longer than the test and the batch closes on max_batch_size."""
    import threading
    import time
    from whisperlive_amd.batching import BatchInferenceWorker, BatchRequest

    sizes, lock = [], threading.Lock()
    started = [threading.Event(), threading.Event()]
    release = threading.Event()

    class W(BatchInferenceWorker):
        def _process_batch(self, batch):
            with lock:
                sizes.append(len(batch))
                k = len(sizes) - 1
            if k < len(started):
                started[k].set()
            assert release.wait(10)
            for r in batch:
                r.result = []
                r.future.set()

    mk = lambda: BatchRequest(audio=np.zeros(1600, np.float32), use_vad=False)
    w = W(MagicMock(max_batch=2), max_batch_size=2, batch_window_ms=20000, lanes=2)
    w.start()
    try:
        a = mk(); t0 = time.monotonic(); w.submit(a)
        assert started[0].wait(5)                           # a runs alone, at once — not after the 20 s window
        assert time.monotonic() - t0 < 2.0 and sizes == [1]
        b, c = mk(), mk(); w.submit(b); w.submit(c)         # lane 0 is held busy: lane 1 collects over its window -> {b, c}
        assert started[1].wait(5) and sizes == [1, 2], sizes
        release.set()
        assert a.future.wait(5) and b.future.wait(5) and c.future.wait(5)
    finally:
        release.set()
        w.stop()


def test_batch_worker_reference_collection_waits_out_the_window_when_idle():
    """wait_when_idle=True is the reference's collection (whisper_live/batch_inference.py:126-153): the first request of a batch
    waits for the window (or for max_batch_size requests) even when nothing is running."""
    import threading
    import time
    from whisperlive_amd.batching import BatchInferenceWorker, BatchRequest

    sizes = []

    class W(BatchInferenceWorker):
        def _process_batch(self, batch):
            sizes.append(len(batch))
            for r in batch:
                r.result = []
                r.future.set()

    mk = lambda: BatchRequest(audio=np.zeros(1600, np.float32), use_vad=False)
    w = W(MagicMock(max_batch=3), max_batch_size=3, batch_window_ms=20000, lanes=1, wait_when_idle=True)
    w.start()
    try:
        reqs = [mk() for _ in range(3)]
        w.submit(reqs[0])
        time.sleep(0.05)
        assert not reqs[0].future.is_set() and sizes == []  # still inside the window: nothing started
        w.submit(reqs[1]); w.submit(reqs[2])                # the batch closes on max_batch_size
        assert all(r.future.wait(5) for r in reqs) and sizes == [3], sizes
    finally:
        w.stop()
