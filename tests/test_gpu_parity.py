"""GPU parity tests (-m gpu): every stage of the HIP hot path, called through the C-ABI, against the CPU oracle
on the same seeded inputs. Tolerances are stated per test:

* log-mel: fp32 kernel vs float64 oracle: max-abs <= 2e-4 over the whole map (values live in roughly [-1, 2]).
* encoder / decoder logits: fp16 MFMA operands with fp32 accumulation and an fp32 residual stream vs the fp32
  oracle evaluated on the SAME fp16-rounded weights: relative RMS <= 2e-3 (encoder states) / 5e-3 (logits),
  max-abs <= 10 x that x ref_rms + 1e-3 — the bounds of tests/test_gpu_full_depth.py.
* search (logits processors + beam/sampling bookkeeping) on injected logits: token-exact, scores to 1e-3.
"""
import numpy as np
import pytest

from tests import helpers as H
from oracle import decoding as odec
from oracle import logmel as olm
from oracle import model as omodel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny(gpu):
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.weights import random_weights
    spec = H.TINY_EN
    w = random_weights(spec, seed=7)
    eng = HipWhisperEngine(spec, w)
    oracle = omodel.WhisperOracle(H.oracle_spec(spec), H.f16_weights(w))
    yield spec, eng, oracle
    eng.close()


@pytest.fixture(scope="module")
def micro128(gpu):
    """128-mel front-end + ragged vocabulary (2310 = 144*16 + 6) + odd layer count."""
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import WhisperSpec
    from whisperlive_amd.weights import random_weights
    spec = WhisperSpec(n_mels=128, d_model=128, n_heads=2, enc_layers=1, dec_layers=3, ffn=512, vocab=2310)
    w = random_weights(spec, seed=3)
    eng = HipWhisperEngine(spec, w)
    oracle = omodel.WhisperOracle(H.oracle_spec(spec), H.f16_weights(w))
    yield spec, eng, oracle
    eng.close()


def _pcm(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1730 * t + 1.0) + 0.05 * rng.standard_normal(n)
    x *= (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t))
    return x.astype(np.float32)


@pytest.mark.parametrize("n", [16000, 17777, 176000, 480000, 41, 200, 721234])
def test_logmel_parity(tiny, n):
    spec, eng, _ = tiny
    slot = eng.create_slot(1, 5)
    try:
        pcm = _pcm(n, n) if n > 400 else (np.random.default_rng(n).standard_normal(n).astype(np.float32) * 0.1)
        T = slot.logmel(pcm)
        got = slot.features()
        ref = olm.log_mel_spectrogram(pcm, spec.n_mels)
        assert T == (n + 160) // 160 and got.shape == ref.shape
        st = H.err_stats(got, ref)
        assert st["max_abs"] <= 2e-4, st
    finally:
        slot.close()


def test_logmel_against_reference_vectors(tiny, micro128):
    """The HIP log-mel against vectors the REFERENCE'S OWN code produced (tests/golden/ref_logmel_golden.npz: outputs of
    whisper_live/transcriber/tensorrt_utils.py::log_mel_spectrogram(padding=160), generated in the build container by
    tests/golden/make_ref_logmel_golden.py — /root/reference does not exist on the GPU box). Tolerance 2e-4 like the oracle
    test (the reference's float32 STFT itself sits up to 6e-5 from the float64 answer, tests/test_reference_logmel_diff.py)."""
    import os
    from whisperlive_amd.synthetic import speech_like_pcm
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_logmel_golden.npz"))
    for i, (sec, seed, n_mels) in enumerate(g["cases"]):
        spec, eng, _ = tiny if int(n_mels) == 80 else micro128
        assert spec.n_mels == int(n_mels)
        slot = eng.create_slot(1, 5)
        try:
            slot.logmel(speech_like_pcm(float(sec), seed=int(seed)))
            got = slot.features()
        finally:
            slot.close()
        want = g[f"logmel_{i}"]
        assert got.shape == want.shape
        st = H.err_stats(got, want)
        assert st["max_abs"] <= 2e-4, (i, st)


def test_logmel_silence_and_128(micro128):
    spec, eng, _ = micro128
    slot = eng.create_slot(1, 5)
    try:
        for pcm in (np.zeros(32000, np.float32), _pcm(48000, 5)):
            slot.logmel(pcm)
            got = slot.features()
            ref = olm.log_mel_spectrogram(pcm, 128)
            assert got.shape == ref.shape == (128, (pcm.size + 160) // 160)
            st = H.err_stats(got, ref)
            assert st["max_abs"] <= 2e-4, st
    finally:
        slot.close()


def test_logmel_ragged_items_one_launch(tiny):
    """The log-mel requests of a slot's items are recorded and launched together in front of the first consumer (engine.hip
    flush_logmel): ragged lengths in one launch (blockIdx.y = item, short items' surplus tiles exit), an audio-buffer growth in the
    middle of the recorded batch, a pending item whose PCM is replaced, and 18 items (> the 16-entry launch table) — each item against
    the oracle AND bit-identical to the same clip computed alone."""
    spec, eng, _ = tiny
    lens = [480000, 41, 17777, 176000, 16000, 200, 300000, 479999, 8000, 160, 161, 4000, 99999, 250000, 31999, 32000, 32001, 123456]
    clips = [(_pcm(n, 100 + i) if n > 400 else np.random.default_rng(n).standard_normal(n).astype(np.float32) * 0.1) for i, n in enumerate(lens)]
    one = eng.create_slot(1, 5)
    slot = eng.create_slot(len(lens), 1)
    try:
        alone = []
        for c in clips:
            one.logmel(c)
            alone.append(one.features().copy())
        order = sorted(range(len(lens)), key=lambda i: lens[i])       # shortest first: the audio buffers grow while requests are recorded
        for i in order:
            assert slot.logmel(clips[i], item=i) == (lens[i] + 160) // 160
        slot.logmel(clips[3], item=2)                                  # item 2 is pending: its PCM is replaced, then requested again
        for i in range(len(lens)):
            want = alone[3] if i == 2 else alone[i]
            got = slot.features(i)
            assert got.shape == want.shape
            assert np.array_equal(got, want), ("item", i, H.err_stats(got, want))
            ref = olm.log_mel_spectrogram(clips[3] if i == 2 else clips[i], spec.n_mels)
            st = H.err_stats(got, ref)
            assert st["max_abs"] <= 2e-4, (i, st)
    finally:
        slot.close()
        one.close()


def _check_close(got, ref, what, rel=2e-3):
    """rel-rms <= 2e-3 (encoder states) / 5e-3 (logits): the bounds of tests/test_gpu_full_depth.py on the same quantities
    (measured 1.8e-4 ... 8e-4); max-abs <= 10 x that bound x the reference's rms + 1e-3. (Until round 6: 2e-2 / 6e-2.)"""
    st = H.err_stats(got, ref)
    ok = st["rel_rms"] <= rel and st["max_abs"] <= 10 * rel * st["ref_rms"] + 1e-3 and np.isfinite(got).all()
    assert ok, (what, st)
    return st


def test_encoder_parity(tiny):
    spec, eng, oracle = tiny
    slot = eng.create_slot(1, 5)
    try:
        pcm = _pcm(11 * 16000, 1)
        T = slot.logmel(pcm)
        feats = slot.features()
        slot.encode(1, seek=[0], seg=[T - 1])
        got = slot.encoder_output(0)
        seg = olm.pad_or_trim(feats[:, : T - 1])
        ref = oracle.encode(seg[None])[0].numpy()
        st = _check_close(got, ref, "encoder")
        print("encoder parity", st)
    finally:
        slot.close()


def test_encoder_from_host_features_and_seek(tiny):
    """encode() on features supplied by the host (the StorageView.from_array path) with a non-zero seek."""
    spec, eng, oracle = tiny
    slot = eng.create_slot(1, 5)
    try:
        rng = np.random.default_rng(11)
        feats = (rng.standard_normal((spec.n_mels, 3700)) * 0.4).astype(np.float32)
        slot.set_features(feats)
        slot.encode(1, seek=[1000], seg=[2700])
        got = slot.encoder_output(0)
        ref = oracle.encode(olm.pad_or_trim(feats[:, 1000:3700])[None])[0].numpy()
        _check_close(got, ref, "encoder(seek)")
    finally:
        slot.close()


@pytest.mark.parametrize("n_tok", [1, 5, 40, 70])
def test_decoder_logits_parity(tiny, n_tok):
    """Teacher-forced decoder (prefill path: 1..4 MFMA row tiles, chunking at 64 rows) vs oracle logits."""
    spec, eng, oracle = tiny
    slot = eng.create_slot(1, 5)
    try:
        pcm = _pcm(5 * 16000, 2)
        T = slot.logmel(pcm)
        feats = slot.features()
        slot.encode(1, seek=[0], seg=[T - 1])
        enc = oracle.encode(olm.pad_or_trim(feats[:, : T - 1])[None])
        rng = np.random.default_rng(n_tok)
        toks = rng.integers(0, spec.vocab, size=n_tok)
        got = slot.debug_decode_logits(toks)
        ref = oracle.decode_logits(enc, toks[None])[0].numpy()
        st = _check_close(got, ref, f"decoder logits n={n_tok}", rel=5e-3)
        print("decoder parity", n_tok, st)
    finally:
        slot.close()


def _injected_case(V, ids, steps, rows, seed, peaky=4.0):
    rng = np.random.default_rng(seed)
    lg = (rng.standard_normal((steps, rows, V)) * peaky).astype(np.float32)
    # make timestamps / eot competitive so every rule fires
    lg[:, :, ids.timestamp_begin:] += 2.0
    lg[:, :, ids.eot] += 6.0
    return lg


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_search_beam_injected(micro128, seed):
    spec, eng, _ = micro128
    ids = H.token_ids_for(spec.vocab)
    slot = eng.create_slot(1, 5)
    try:
        steps, B = 40, 5
        lg = _injected_case(spec.vocab, ids, steps, B, seed)
        prompt = [ids.sot] if seed % 2 == 0 else [ids.sot - 1, 17, 29, ids.sot]
        kw = dict(beam_size=B, patience=1.0 if seed < 3 else 2.0, max_length=len(prompt) + 30,
                  suppress_tokens=H.default_suppress(ids), length_penalty=1.0)
        o = odec.GenOptions(ids=ids, **kw)
        ref = odec.generate(odec.InjectedLogits(lg), prompt, o)
        got = slot.debug_search(lg, prompt, H.engine_ids(ids), **kw)
        assert got.sequences_ids[0] == ref.sequences_ids[0], (got.sequences_ids, ref.sequences_ids)
        assert abs(got.scores[0] - ref.scores[0]) <= 1e-3 * max(1.0, abs(ref.scores[0]))
    finally:
        slot.close()


def test_search_options_injected(micro128):
    """repetition penalty, no-repeat-ngram, no timestamps prompt, suppress_blank off, length_penalty 0 (early exit)."""
    spec, eng, _ = micro128
    ids = H.token_ids_for(spec.vocab)
    slot = eng.create_slot(1, 5)
    try:
        lg = _injected_case(spec.vocab, ids, 30, 5, 9, peaky=2.0)
        cases = [
            dict(prompt=[ids.sot, ids.no_timestamps], kw=dict(beam_size=5, max_length=25, repetition_penalty=1.3)),
            dict(prompt=[ids.sot], kw=dict(beam_size=3, max_length=20, no_repeat_ngram_size=2, suppress_blank=False)),
            dict(prompt=[ids.sot], kw=dict(beam_size=4, max_length=28, length_penalty=0.0, num_hypotheses=2)),
            dict(prompt=[ids.sot], kw=dict(beam_size=1, max_length=20, sampling_temperature=0.0)),   # greedy
        ]
        for c in cases:
            o = odec.GenOptions(ids=ids, suppress_tokens=H.default_suppress(ids), **c["kw"])
            rows = o.beam_size if o.beam_size > 1 else max(1, o.num_hypotheses)
            ref = odec.generate(odec.InjectedLogits(lg[:, :rows]), c["prompt"], o)
            got = slot.debug_search(np.ascontiguousarray(lg[:, :rows]), c["prompt"], H.engine_ids(ids),
                                    suppress_tokens=H.default_suppress(ids), **c["kw"])
            assert got.sequences_ids == ref.sequences_ids, (c, got.sequences_ids, ref.sequences_ids)
            np.testing.assert_allclose(got.scores, ref.scores, rtol=1e-3, atol=1e-3)
    finally:
        slot.close()


def test_search_sampling_injected(micro128):
    spec, eng, _ = micro128
    ids = H.token_ids_for(spec.vocab)
    slot = eng.create_slot(1, 5)
    try:
        lg = _injected_case(spec.vocab, ids, 24, 5, 21, peaky=3.0)
        kw = dict(beam_size=1, num_hypotheses=5, sampling_temperature=0.6, sampling_topk=0, max_length=21, seed=1234,
                  suppress_tokens=H.default_suppress(ids))
        ref = odec.generate(odec.InjectedLogits(lg), [ids.sot], odec.GenOptions(ids=ids, **kw))
        got = slot.debug_search(lg, [ids.sot], H.engine_ids(ids), **kw)
        got2 = slot.debug_search(lg, [ids.sot], H.engine_ids(ids), **kw)
        assert got.sequences_ids == got2.sequences_ids          # same seed -> same draw
        assert sorted(map(tuple, got.sequences_ids)) == sorted(map(tuple, ref.sequences_ids))
        np.testing.assert_allclose(sorted(got.scores), sorted(ref.scores), rtol=2e-3, atol=2e-3)
        kw["seed"] = 99
        other = slot.debug_search(lg, [ids.sot], H.engine_ids(ids), **kw)
        assert other.sequences_ids != got.sequences_ids
    finally:
        slot.close()


def _gen_both(spec, eng, oracle, pcm, prompt, ids, **kw):
    slot = eng.create_slot(1, 5)
    try:
        T = slot.logmel(pcm)
        feats = slot.features()
        slot.encode(1, seek=[0], seg=[T - 1])
        got = slot.generate([prompt], H.engine_ids(ids), **kw)[0]
        enc = oracle.encode(olm.pad_or_trim(feats[:, : T - 1])[None])
        ref = odec.generate(H.NetProvider(oracle, enc), prompt, odec.GenOptions(ids=ids, **kw))
        return got, ref, slot.timings()
    finally:
        slot.close()


def _margin_tolerant_equal(got, ref):
    """fp16-vs-fp32 logits may flip a near-tie; require equal tokens up to the first position, then
    require that at least the first 8 tokens agree (random weights give flat-ish distributions)."""
    g, r = got.sequences_ids[0], ref.sequences_ids[0]
    n = 0
    while n < min(len(g), len(r)) and g[n] == r[n]:
        n += 1
    return n


def test_generate_beam_end_to_end(tiny):
    spec, eng, oracle = tiny
    ids = H.token_ids_for(spec.vocab)
    kw = dict(beam_size=5, patience=1.0, max_length=1 + 24, suppress_tokens=H.default_suppress(ids))
    got, ref, tm = _gen_both(spec, eng, oracle, _pcm(6 * 16000, 4), [ids.sot], ids, **kw)
    n = _margin_tolerant_equal(got, ref)
    print("beam e2e common prefix", n, len(ref.sequences_ids[0]), got.scores, ref.scores, tm)
    assert n >= min(8, len(ref.sequences_ids[0])), (got.sequences_ids, ref.sequences_ids)
    assert abs(got.no_speech_prob - ref.no_speech_prob) <= 5e-3 + 0.05 * ref.no_speech_prob


def test_generate_long_prompt_and_multilingual_style_prompt(tiny):
    """prompt with previous-text conditioning (prefill > 64 tokens, sot in the middle -> no_speech from prefill)."""
    spec, eng, oracle = tiny
    ids = H.token_ids_for(spec.vocab)
    rng = np.random.default_rng(5)
    prev = rng.integers(300, 20000, size=90).tolist()
    prompt = [ids.sot - 1] + prev + [ids.sot, ids.sot + 1, ids.sot + 2]
    kw = dict(beam_size=5, max_length=len(prompt) + 12, suppress_tokens=H.default_suppress(ids))
    got, ref, _ = _gen_both(spec, eng, oracle, _pcm(4 * 16000, 6), prompt, ids, **kw)
    n = _margin_tolerant_equal(got, ref)
    assert n >= min(6, len(ref.sequences_ids[0])), (got.sequences_ids, ref.sequences_ids)
    assert abs(got.no_speech_prob - ref.no_speech_prob) <= 5e-3 + 0.05 * ref.no_speech_prob


def test_detect_language_parity(tiny):
    spec, eng, oracle = tiny
    ids = H.token_ids_for(spec.vocab)
    slot = eng.create_slot(1, 5)
    try:
        T = slot.logmel(_pcm(3 * 16000, 8))
        feats = slot.features()
        slot.encode(1, seek=[0], seg=[T - 1])
        lang = list(range(ids.sot + 1, ids.sot + 100))
        got = slot.detect_language(1, ids.sot, lang)[0]
        enc = oracle.encode(olm.pad_or_trim(feats[:, : T - 1])[None])
        lg = oracle.decode_logits(enc, np.asarray([[ids.sot]]))[0, 0].numpy()[lang]
        ref = np.exp(lg - lg.max()); ref /= ref.sum()
        assert abs(got.sum() - 1.0) < 1e-4
        np.testing.assert_allclose(got, ref, atol=2e-3)
    finally:
        slot.close()


def test_batched_equals_single(tiny):
    """batch of 2 different clips through ONE encode + ONE generate == each clip alone (batch_inference path)."""
    spec, eng, oracle = tiny
    ids = H.token_ids_for(spec.vocab)
    kw = dict(beam_size=5, max_length=1 + 16, suppress_tokens=H.default_suppress(ids))
    clips = [_pcm(7 * 16000, 31), _pcm(9 * 16000, 32)]
    singles = []
    for c in clips:
        s1 = eng.create_slot(1, 5)
        T = s1.logmel(c); s1.encode(1, seek=[0], seg=[T - 1])
        singles.append((s1.generate([[ids.sot]], H.engine_ids(ids), **kw)[0], s1.encoder_output(0)))
        s1.close()
    sb = eng.create_slot(2, 5)
    try:
        Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
        sb.encode(2, seek=[0, 0], seg=[t - 1 for t in Ts])
        res = sb.generate([[ids.sot], [ids.sot]], H.engine_ids(ids), **kw)
        for i in range(2):
            np.testing.assert_allclose(sb.encoder_output(i), singles[i][1], atol=2e-3, rtol=0)
            assert res[i].sequences_ids == singles[i][0].sequences_ids
            assert abs(res[i].scores[0] - singles[i][0].scores[0]) < 1e-3
    finally:
        sb.close()


def test_generate_is_deterministic_and_graph_equals_eager(tiny, monkeypatch):
    spec, eng, _ = tiny
    ids = H.token_ids_for(spec.vocab)
    slot = eng.create_slot(1, 5)
    try:
        T = slot.logmel(_pcm(5 * 16000, 77)); slot.encode(1, seek=[0], seg=[T - 1])
        kw = dict(beam_size=5, max_length=30, suppress_tokens=H.default_suppress(ids))
        a = slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0]
        b = slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0]
        assert a.sequences_ids == b.sequences_ids and a.scores == b.scores
    finally:
        slot.close()


@pytest.mark.parametrize("n_text", [9, 90])
def test_align_parity(tiny, n_text):
    """wlx_align (cross-attention scores of the alignment heads on the device, softmax / normalise / median / DTW on
    the host side of the call) vs oracle/alignment.py: token probabilities to 2e-3 + 2 %; the DTW path must be a valid
    monotone path whose cost ON THE ORACLE'S MATRIX is within 0.5 % of the optimum (fp16 attention can move a step,
    DTW is discontinuous) and mostly identical."""
    from oracle import alignment as oal
    spec, eng, oracle = tiny
    ids = H.token_ids_for(spec.vocab)
    slot = eng.create_slot(1, 5)
    try:
        pcm = _pcm(8 * 16000, 12)
        T = slot.logmel(pcm)
        feats = slot.features()
        slot.encode(1, seek=[0], seg=[T - 1])
        enc = oracle.encode(olm.pad_or_trim(feats[:, : T - 1])[None])
        rng = np.random.default_rng(n_text)
        text = rng.integers(300, ids.eot - 1, size=n_text).tolist()
        sot_seq = [ids.sot]
        heads = [(l, h) for l in range(spec.dec_layers // 2, spec.dec_layers) for h in range(spec.n_heads)]
        tokens = sot_seq + [ids.no_timestamps] + text + [ids.eot]
        ti, fi, probs = slot.align(tokens, len(sot_seq), T - 1, heads, ids.eot, median_filter_width=7)
        rti, rfi, rprobs, matrix = oal.align(oracle, enc, sot_seq, ids.no_timestamps, text, ids.eot, T - 1, heads, 7)
        np.testing.assert_allclose(probs, rprobs, atol=2e-3, rtol=2e-2)
        N, M = matrix.shape
        assert ti[0] == 0 and fi[0] == 0 and ti[-1] == N - 1 and fi[-1] == M - 1
        dt, df = np.diff(ti), np.diff(fi)
        assert ((dt == 0) | (dt == 1)).all() and ((df == 0) | (df == 1)).all() and ((dt + df) >= 1).all()
        cost = lambda a, b: float((-matrix)[a, b].sum())
        c_got, c_ref = cost(ti, fi), cost(rti, rfi)
        assert c_got <= c_ref + 5e-3 * abs(c_ref) + 1e-3, (c_got, c_ref)
        same = len(set(zip(ti.tolist(), fi.tolist())) & set(zip(rti.tolist(), rfi.tolist()))) / len(rti)
        print("align", n_text, "path overlap", same, "cost", c_got, c_ref)
        # "mostly identical" only means something where the optimum is well separated: on these seeded random weights the
        # attention matrix is nearly flat, two monotone paths that share half their cells can differ by 2e-4 of the cost
        # (seen when the encoder attention kernel changed its summation order), and DTW then picks either. A path that is
        # as good as the oracle's ON THE ORACLE'S MATRIX to 0.1 % passes whatever its overlap.
        assert same >= 0.8 or c_got <= c_ref + 1e-3 * abs(c_ref), (same, c_got, c_ref)
    finally:
        slot.close()
