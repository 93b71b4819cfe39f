"""CPU tests of the host logic around the five engine calls: tokenizer glue, VAD segmentation / time map, prompt
construction, timestamp splitting, temperature fallback, the seek loop, language detection. The engine is scripted
(tests/fakes.py) — numerics are covered by the GPU parity tests."""
import numpy as np
import pytest

from tests.fakes import FakeEngine
from whisperlive_amd import vad
from whisperlive_amd.engine import GenerationResult
from whisperlive_amd.tokenizer import LANGUAGE_CODES, Tokenizer, synthetic_tokenizer
from whisperlive_amd.transcriber import (WhisperModelHIP, get_compression_ratio, get_suppressed_tokens, pad_or_trim,
                                         restore_speech_timestamps)
from whisperlive_amd.types import Segment, TranscriptionOptions

V = 2310


@pytest.fixture()
def model():
    eng = FakeEngine()
    return WhisperModelHIP("fake", engine=eng, hf_tokenizer=synthetic_tokenizer(V), vad_model=vad.EnergyGateModel()), eng


@pytest.fixture()
def ml_model():
    eng = FakeEngine()
    return WhisperModelHIP("fake", engine=eng, hf_tokenizer=synthetic_tokenizer(V), multilingual=True), eng


# ------------------------------------------------------------------------------------------------ tokenizer
def test_tokenizer_special_ids_by_name_real_layouts():
    en = Tokenizer(synthetic_tokenizer(51864), False)
    assert (en.eot, en.sot, en.translate, en.transcribe, en.sot_lm, en.sot_prev, en.no_speech, en.no_timestamps,
            en.timestamp_begin) == (50256, 50257, 50357, 50358, 50359, 50360, 50361, 50362, 50363)
    assert en.sot_sequence == [50257] and en.language_code == "en"
    ml = Tokenizer(synthetic_tokenizer(51865), True, task="translate", language="de")
    assert ml.sot_sequence == [50258, 50258 + 1 + LANGUAGE_CODES.index("de"), ml.translate] and ml.timestamp_begin == 50364
    v3 = Tokenizer(synthetic_tokenizer(51866), True, task="transcribe", language="yue")
    assert v3.timestamp_begin == 50365 and v3.language == 50258 + 100
    with pytest.raises(ValueError):
        Tokenizer(synthetic_tokenizer(51865), True, task="summarize", language="en")
    with pytest.raises(ValueError):
        Tokenizer(synthetic_tokenizer(51865), True, task="transcribe", language="xx")


def test_tokenizer_encode_decode_and_suppression_list():
    tk = Tokenizer(synthetic_tokenizer(V), False)
    ids = tk.encode(" hello (world)")
    assert tk.decode(ids + [tk.eot, tk.timestamp_begin + 3]) == " hello (world)"      # specials are dropped
    assert tk.decode_with_timestamps([tk.timestamp_begin] + ids + [tk.timestamp_begin + 50]).startswith("<|0.00|> hello")
    ns = tk.non_speech_tokens
    assert tk.encode("(")[0] in ns and tk.encode(" -")[0] in ns and tk.encode("a")[0] not in ns
    sup = get_suppressed_tokens(tk, [-1])
    assert set(ns) <= set(sup) and {tk.transcribe, tk.translate, tk.sot, tk.sot_prev, tk.sot_lm} <= set(sup)
    assert get_suppressed_tokens(tk, [5, 9]) == tuple(sorted({5, 9, tk.transcribe, tk.translate, tk.sot, tk.sot_prev, tk.sot_lm}))
    assert list(sup) == sorted(set(sup))


# ------------------------------------------------------------------------------------------------ VAD
def test_vad_hysteresis_padding_and_map():
    opt = vad.VadOptions(threshold=0.5)                  # min_silence 2000 ms, pad 400 ms (reference default)
    n_win = 400                                          # 12.8 s
    probs = np.zeros(n_win, np.float32)
    probs[50:120] = 0.9                                  # speech A: 1.6 s .. 3.84 s
    probs[125:130] = 0.4                                 # between thresholds (0.35 .. 0.5): does not end speech
    probs[130:150] = 0.9
    probs[260:300] = 0.9                                 # speech B after > 2 s silence
    n = n_win * 512
    segs = vad.speech_segments_from_probs(probs, n, opt)
    assert len(segs) == 2
    assert segs[0]["start"] == 50 * 512 - 6400 and segs[0]["end"] == 150 * 512 + 6400
    assert segs[1]["start"] == 260 * 512 - 6400 and segs[1]["end"] == 300 * 512 + 6400
    audio = np.arange(n, dtype=np.float32)
    chunks, meta = vad.collect_chunks(audio, segs)
    assert len(chunks) == 1 and chunks[0].shape[0] == sum(s["end"] - s["start"] for s in segs)
    assert chunks[0][0] == segs[0]["start"]
    m = vad.SpeechTimestampsMap(segs, 16000)
    assert m.get_original_time(0.0) == round(segs[0]["start"] / 16000, 2)
    t_in_b = (segs[0]["end"] - segs[0]["start"]) / 16000 + 0.5
    assert m.get_original_time(t_in_b) == round(segs[1]["start"] / 16000 + 0.5, 2)
    assert m.get_chunk_index(1e9) == 1


def test_vad_close_segments_share_the_gap_and_silence_is_empty():
    opt = vad.VadOptions(threshold=0.5, min_silence_duration_ms=100, speech_pad_ms=400)
    probs = np.zeros(200, np.float32)
    probs[10:40] = 0.9
    probs[50:90] = 0.9                                   # gap of 10 windows = 5120 samples < 2 * 6400
    segs = vad.speech_segments_from_probs(probs, 200 * 512, opt)
    assert len(segs) == 2 and segs[0]["end"] == segs[1]["start"] == 40 * 512 + 5120 // 2
    assert vad.speech_segments_from_probs(np.zeros(50, np.float32), 50 * 512, opt) == []
    chunks, _ = vad.collect_chunks(np.ones(100, np.float32), [])
    assert len(chunks) == 1 and chunks[0].shape[0] == 0
    # speech running to the end of the audio is closed at n_samples
    probs = np.zeros(100, np.float32); probs[60:] = 0.9
    segs = vad.speech_segments_from_probs(probs, 100 * 512 - 100, vad.VadOptions())
    assert segs[-1]["end"] == 100 * 512 - 100
    # max_speech_duration splits long speech at the last >98 ms silence
    opt = vad.VadOptions(max_speech_duration_s=4.0, min_silence_duration_ms=2000, speech_pad_ms=0)
    probs = np.full(400, 0.9, np.float32); probs[100:110] = 0.1
    segs = vad.speech_segments_from_probs(probs, 400 * 512, opt)
    assert len(segs) >= 2 and segs[0]["end"] == 100 * 512


def test_vad_energy_gate_is_a_labelled_stand_in():
    x = np.zeros(16000, np.float32)
    assert vad.get_speech_timestamps(x, vad.VadOptions(), model=vad.EnergyGateModel()) == []
    t = np.arange(32000) / 16000.0
    y = np.concatenate([np.zeros(16000), 0.3 * np.sin(2 * np.pi * 300 * t), np.zeros(16000)]).astype(np.float32)
    segs = vad.get_speech_timestamps(y, vad.VadOptions(min_silence_duration_ms=300, speech_pad_ms=0), model=vad.EnergyGateModel())
    assert len(segs) == 1 and abs(segs[0]["start"] - 16000) <= 512 and abs(segs[0]["end"] - 48000) <= 1024


# ------------------------------------------------------------------------------------------------ helpers
def test_small_helpers():
    a = np.ones((2, 10), np.float32)
    assert pad_or_trim(a, 16).shape == (2, 16) and pad_or_trim(a, 16)[:, 10:].sum() == 0 and pad_or_trim(a, 4).shape == (2, 4)
    assert get_compression_ratio("ab" * 200) > 2.4 > get_compression_ratio("the quick brown fox")
    segs = [Segment(1, 0, 0.5, 1.0, "x", [1], 0, 0, 0)]
    out = restore_speech_timestamps(segs, [{"start": 32000, "end": 64000}], 16000)
    assert (out[0].start, out[0].end) == (2.5, 3.0)


# ------------------------------------------------------------------------------------------------ prompt / split
def test_get_prompt_shapes(model):
    m, _ = model
    tk = Tokenizer(m.hf_tokenizer, False)
    assert m.get_prompt(tk, []) == [tk.sot]
    assert m.get_prompt(tk, [], without_timestamps=True) == [tk.sot, tk.no_timestamps]
    prev = list(range(300, 900))
    p = m.get_prompt(tk, prev)
    assert p[0] == tk.sot_prev and p[1:-1] == prev[-223:] and p[-1] == tk.sot and len(p) == 225
    hw = m.get_prompt(tk, [], hotwords="alpha beta")
    assert hw[0] == tk.sot_prev and hw[-1] == tk.sot and hw[1:-1] == tk.encode(" alpha beta")
    px = m.get_prompt(tk, [], prefix="hello", hotwords="ignored when prefix is set")
    assert px == [tk.sot, tk.timestamp_begin] + tk.encode(" hello")
    mlt = Tokenizer(synthetic_tokenizer(V), True, task="transcribe", language="fr")
    assert m.get_prompt(mlt, [7])[:2] == [mlt.sot_prev, 7] and m.get_prompt(mlt, [7])[2:] == mlt.sot_sequence


def test_split_segments_by_timestamps(model):
    m, _ = model
    tk = Tokenizer(m.hf_tokenizer, False)
    tb = tk.timestamp_begin
    a, b = tk.encode(" ab"), tk.encode(" cd")
    # two closed segments, then an unfinished one -> seek to the last closed timestamp
    toks = [tb + 0] + a + [tb + 100, tb + 100] + b + [tb + 250, tb + 250] + a
    segs, seek, single = m._split_segments_by_timestamps(tk, toks, 10.0, 3000, 30.0, 1000)
    assert [(s["start"], s["end"]) for s in segs] == [(10.0, 12.0), (12.0, 15.0)] and seek == 1000 + 250 * 2 and not single
    # single trailing timestamp -> whole window consumed
    toks = [tb + 0] + a + [tb + 100, tb + 100] + b + [tb + 200]
    segs, seek, single = m._split_segments_by_timestamps(tk, toks, 0.0, 2500, 25.0, 0)
    assert single and seek == 2500 and len(segs) == 2 and segs[1]["end"] == 4.0
    # no consecutive timestamps: one segment, duration from the last timestamp if any
    segs, seek, _ = m._split_segments_by_timestamps(tk, [tb] + a + [tb + 77], 3.0, 1200, 12.0, 5)
    assert len(segs) == 1 and segs[0]["start"] == 3.0 and abs(segs[0]["end"] - (3.0 + 1.54)) < 1e-9 and seek == 1205
    segs, _, _ = m._split_segments_by_timestamps(tk, a, 3.0, 1200, 12.0, 5)
    assert segs[0]["end"] == 15.0


# ------------------------------------------------------------------------------------------------ fallback
def _res(tokens, score, nsp=0.01):
    return lambda prompts, ids, kw: [GenerationResult([list(tokens)], [score], nsp)]


def test_generate_with_fallback_accepts_first_good_result(model):
    m, eng = model
    tk = Tokenizer(m.hf_tokenizer, False)
    good = tk.encode(" fine words here")
    eng.generate_script = [_res(good, -0.2)]
    enc = m.encode(np.zeros((80, 3000), np.float32))
    r, avg, temp, cr = m.generate_with_fallback(enc, [tk.sot], tk, TranscriptionOptions(suppress_tokens=[1]))
    assert r.sequences_ids[0] == good and temp == 0.0 and abs(avg - (-0.2 * len(good)) / (len(good) + 1)) < 1e-9
    kw = eng.slots[0].calls[-1][2]
    assert kw["beam_size"] == 5 and kw["sampling_temperature"] == 0.0 and kw["max_length"] == 448
    assert kw["max_initial_timestamp_index"] == 50 and kw["suppress_tokens"] == [1]


def test_generate_with_fallback_walks_temperatures_and_picks_best(model):
    m, eng = model
    tk = Tokenizer(m.hf_tokenizer, False)
    rep = tk.encode(" la" * 120)                        # compression ratio > 2.4
    low = tk.encode(" unlikely text")
    eng.generate_script = [_res(rep, -0.1), _res(low, -3.0), _res(low, -2.0)] + [_res(low, -2.5)] * 3
    enc = m.encode(np.zeros((80, 3000), np.float32))
    r, avg, temp, cr = m.generate_with_fallback(enc, [tk.sot], tk, TranscriptionOptions())
    gens = [c for c in eng.slots[0].calls if c[0] == "generate"]
    assert len(gens) == 6 and temp == 1.0                 # all failed: last temperature is reported
    assert abs(avg - (-2.0 * len(low)) / (len(low) + 1)) < 1e-9      # best avg_logprob among those under the CR threshold
    assert gens[1][2]["beam_size"] == 1 and gens[1][2]["num_hypotheses"] == 5 and abs(gens[1][2]["sampling_temperature"] - 0.2) < 1e-9
    # silence short-circuits the fallback
    eng.generate_script = [_res(low, -3.0, nsp=0.9)]
    r, avg, temp, _ = m.generate_with_fallback(enc, [tk.sot], tk, TranscriptionOptions())
    assert temp == 0.0
    with pytest.raises(ValueError):
        m.generate_with_fallback(enc, [tk.sot] * 10, tk, TranscriptionOptions(max_new_tokens=440))


# ------------------------------------------------------------------------------------------------ transcribe / seek loop
def test_transcribe_single_window_and_empty_after_vad(model):
    m, eng = model
    tk = Tokenizer(m.hf_tokenizer, False)
    tb = tk.timestamp_begin
    a, b = tk.encode(" first part"), tk.encode(" second")
    eng.default_tokens = [tb] + a + [tb + 120, tb + 120] + b + [tb + 300]
    audio = (np.random.default_rng(0).standard_normal(6 * 16000) * 0.1).astype(np.float32)
    segs, info = m.transcribe(audio, language="en", vad_filter=False)
    assert [s.text for s in segs] == [" first part", " second"] and (segs[0].start, segs[0].end, segs[1].end) == (0.0, 2.4, 6.0)
    assert info.language == "en" and info.duration == 6.0 and segs[0].id == 1 and segs[0].words is None
    calls = eng.slots[0].calls
    assert calls[0] == ("logmel", 0, 96000) and calls[1] == ("encode", 1, [0], [600])       # content frames = T-1
    # VAD removing everything -> (None, None)
    silent = np.zeros(3 * 16000, np.float32)
    assert m.transcribe(silent, vad_filter=True, vad_parameters={"threshold": 0.5}) == (None, None)
    with pytest.raises(TypeError):
        m.transcribe(12345)
    with pytest.raises(FileNotFoundError):
        m.transcribe("/nonexistent/file.wav")


def test_transcribe_long_audio_seeks_and_conditions_on_previous_text(model):
    m, eng = model
    tk = Tokenizer(m.hf_tokenizer, False)
    tb = tk.timestamp_begin
    a = tk.encode(" chunk")
    # window 1 ends with a closed pair at 20 s and an unfinished tail -> seek = 2000; window 2 closes at its end
    eng.generate_script = [
        lambda p, i, k: [GenerationResult([[tb] + a + [tb + 1000, tb + 1000] + a], [-0.1], 0.0)],
        lambda p, i, k: [GenerationResult([[tb] + a + [tb + 750]], [-0.1], 0.0)],
    ]
    audio = np.zeros(35 * 16000, np.float32) + 0.01
    segs, _ = m.transcribe(audio, language="en")
    enc_calls = [c for c in eng.slots[0].calls if c[0] == "encode"]
    assert enc_calls[0][2:] == ([0], [3000]) and enc_calls[1][2:] == ([2000], [1500])
    assert [(s.start, s.end) for s in segs] == [(0.0, 20.0), (20.0, 35.0)] and segs[1].seek == 2000
    gens = [c for c in eng.slots[0].calls if c[0] == "generate"]
    assert gens[0][1] == [[tk.sot]] and gens[1][1] == [[tk.sot_prev] + [tb] + a + [tb + 1000] + [tk.sot]]


def test_no_speech_skip_and_prompt_reset(model):
    m, eng = model
    tk = Tokenizer(m.hf_tokenizer, False)
    tb = tk.timestamp_begin
    a = tk.encode(" x")
    eng.generate_script = [lambda p, i, k: [GenerationResult([[tb] + a + [tb + 100]], [-5.0], 0.95)]]
    segs, _ = m.transcribe(np.zeros(5 * 16000, np.float32) + 0.01, language="en")
    assert segs == []                                      # no_speech_prob > 0.6 and avg_logprob < -1 -> skipped


def test_language_detection_and_multilingual_prompt(ml_model):
    m, eng = ml_model
    eng.lang_index = LANGUAGE_CODES.index("fr")
    tk = Tokenizer(m.hf_tokenizer, True, task="transcribe", language="fr")
    eng.default_tokens = [tk.timestamp_begin] + tk.encode(" bonjour") + [tk.timestamp_begin + 50]
    segs, info = m.transcribe(np.zeros(2 * 16000, np.float32) + 0.01)
    assert info.language == "fr" and abs(info.language_probability - 0.8) < 1e-6 and info.all_language_probs[0][0] == "fr"
    gen = [c for c in eng.slots[0].calls if c[0] == "generate"][0]
    assert gen[1] == [tk.sot_sequence]
    res = m.model.detect_language(m.encode(np.zeros((80, 3000), np.float32)))
    assert res[0][0][0] == "<|fr|>" and len(res[0]) == 99 and res[0][0][1] > res[0][1][1]
    en_only = WhisperModelHIP("fake", engine=FakeEngine(), hf_tokenizer=synthetic_tokenizer(V))
    s, i = en_only.transcribe(np.zeros(16000, np.float32) + 0.01, language="de")
    assert i.language == "en"                              # English-only model overrides the requested language


def test_model_construction_errors():
    with pytest.raises(ValueError):
        WhisperModelHIP("fake", device="cpu", engine=FakeEngine(), hf_tokenizer=synthetic_tokenizer(V))
    with pytest.raises(ValueError):
        WhisperModelHIP("fake", engine=FakeEngine(), hf_tokenizer=synthetic_tokenizer(V + 16))
    with pytest.raises(FileNotFoundError):
        WhisperModelHIP("/nonexistent/model/dir")


def test_tokenizer_json_without_timestamp_entries_is_accepted():
    """Older converted checkpoints (Systran faster-whisper-tiny … large-v2) ship a tokenizer.json that ends at
    <|notimestamps|>: ~50364 ids against a 51865-row model. faster-whisper derives timestamp ids as no_timestamps + 1 and
    never looks them up, so such a tokenizer must load and transcribe."""
    tok = synthetic_tokenizer(V, timestamps=False)
    assert tok.get_vocab_size() == V - 1501 and tok.token_to_id("<|0.00|>") is None
    eng = FakeEngine()
    m = WhisperModelHIP("fake", engine=eng, hf_tokenizer=tok)
    tb = m.token_ids.timestamp_begin
    assert tb == V - 1501 == Tokenizer(tok, False).no_timestamps + 1
    words = Tokenizer(tok, False).encode(" hello there")
    eng.default_tokens = [tb] + words + [tb + 100]
    segs, _info = m.transcribe(np.zeros(3 * 16000, np.float32) + 0.01, language="en")
    assert [s.text for s in segs] == [" hello there"] and (segs[0].start, segs[0].end) == (0.0, 2.0)
    res = m.model.generate(m.encode(np.zeros((80, 3000), np.float32)), [[m.token_ids.sot]])
    assert res[0].sequences[0][0] == "<|0.00|>" and res[0].sequences[0][-1] == "<|2.00|>"
    # specials that do not fit the model's vocabulary are still refused
    small = FakeEngine()
    small.spec = type(small.spec)(80, 128, 2, 1, 1, 512, V - 1501 + 10)
    with pytest.raises(ValueError, match="do not fit"):
        WhisperModelHIP("fake", engine=small, hf_tokenizer=tok)


def test_metrics_record_is_bounded_and_xrt_stays_exact():
    from whisperlive_amd import metrics
    metrics.snapshot(reset=True)
    n = metrics.WINDOW + 1000
    for i in range(n):
        metrics.track_transcription_latency(0.01 if i < n - 100 else 1.0)
        metrics.track_audio_processed(0.5)
    assert len(metrics._latencies) == metrics.WINDOW                       # a long-running server does not grow
    snap = metrics.snapshot(reset=True)
    assert snap["chunks"] == n and abs(snap["audio_s"] - 0.5 * n) < 1e-6
    assert abs(snap["xrt"] - (0.5 * n) / (0.01 * (n - 100) + 100.0)) < 1e-9   # sums cover every chunk, not the window
    assert snap["p50_latency_s"] == 0.01 and snap["p95_latency_s"] == 0.01
    assert metrics.snapshot()["chunks"] == 0
