"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors (tests/golden/hf_golden.*), which were
produced by Hugging Face transformers — the independent implementation of the published Whisper algorithm that is
importable offline (tests/golden/make_golden.py is the generating script; transformers is NOT needed here).
The reference's own tests pin no tensor on this path."""
import json
import os

import numpy as np
import pytest

from oracle import decoding as odec
from oracle import logmel as olm
from oracle import model as omodel
from oracle.provider import NetProvider
from tests.golden import make_golden as mg

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    z = np.load(os.path.join(HERE, "golden", "hf_golden.npz"))
    with open(os.path.join(HERE, "golden", "hf_golden.json")) as f:
        meta = json.load(f)
    return z, meta


@pytest.mark.parametrize("n_mels", [80, 128])
def test_mel_filterbank_equals_hf(golden, n_mels):
    z, _ = golden
    ours = olm.mel_filters(n_mels)
    np.testing.assert_allclose(ours, z[f"mel_filters_{n_mels}"], rtol=0, atol=4e-9)   # <= 1 float32 ulp
    nz = int((ours != 0).sum())
    assert nz == (391 if n_mels == 80 else 394)      # sparsity the HIP mel reduction exploits


@pytest.mark.parametrize("n_mels", [80, 128])
def test_logmel_equals_hf_feature_extractor(golden, n_mels):
    z, meta = golden
    pcm = olm.speech_like_pcm(meta["logmel"]["seconds"], seed=meta["logmel"]["seed"])
    ref = z[f"logmel_{n_mels}"]
    for precise, tol in ((False, 2e-5), (True, 2e-4)):
        got = olm.log_mel_spectrogram(pcm, n_mels, padding=0, precise=precise)
        assert got.shape == ref.shape == (n_mels, pcm.size // 160)
        assert np.abs(got - ref).max() <= tol


def test_logmel_faster_whisper_shape_contract():
    """padding=160 then drop the last STFT frame: T = (n + 160) // 160 (30 s -> 3001), SURVEY.md §8a row 8."""
    for n in (16000, 17777, 480000, 41, 1):
        x = np.random.default_rng(n).standard_normal(n).astype(np.float32) * 0.1
        assert olm.log_mel_spectrogram(x, 80).shape == (80, (n + 160) // 160)
    f = olm.log_mel_spectrogram(np.zeros(16000, np.float32), 80)
    assert np.allclose(f, (np.log10(1e-10) + 4.0) / 4.0)       # silence: log floor, no clamp effect
    assert olm.pad_or_trim(f).shape == (80, 3000) and np.all(olm.pad_or_trim(f)[:, 101:] == 0.0)
    assert olm.pad_or_trim(np.ones((2, 3100), np.float32)).shape == (2, 3000)


@pytest.mark.parametrize("case", ["m64", "m128"])
def test_network_equals_hf(golden, case):
    z, _ = golden
    spec, w = mg.np_weights(case)
    oracle = omodel.WhisperOracle(omodel.Spec(spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers, spec.dec_layers,
                                              spec.ffn, spec.vocab), w)
    feats = mg.case_features(case)
    enc = oracle.encode(feats)
    np.testing.assert_allclose(enc[0, ::25].numpy(), z[f"{case}_enc_rows"], rtol=0, atol=2e-4)
    toks = mg.case_tokens(case)
    logits = oracle.decode_logits(enc, toks[None])[0].numpy()
    np.testing.assert_allclose(logits, z[f"{case}_logits"], rtol=0, atol=5e-4)
    # the incremental (KV-cached) decoder the search uses == the teacher-forced one
    sd = omodel.StepDecoder(oracle, enc)
    inc = np.concatenate([sd.step(toks[None, :4])[0].numpy(), sd.step(toks[None, 4:5])[0].numpy(),
                          sd.step(toks[None, 5:])[0].numpy()])
    np.testing.assert_allclose(inc, logits, rtol=0, atol=2e-4)


def test_timestamp_rules_equal_hf_processor(golden):
    z, meta = golden
    vocab, L = meta["ts"]["vocab"], meta["ts"]["layout"]
    masks = np.unpackbits(z["ts_masks"], axis=1)[:, :vocab].astype(bool)
    ids = odec.TokenIds(**L)
    o = odec.GenOptions(ids=ids, suppress_blank=False, suppress_tokens=(), max_initial_timestamp_index=50)
    cases = mg.ts_cases(vocab)
    assert len(cases) == meta["ts"]["n_cases"] == masks.shape[0]
    for (hist, seed, peak), ref in zip(cases, masks):
        v, lse, mx = odec.process_logits(mg.ts_scores(vocab, seed, peak), hist, o, apply_ts=True)
        assert np.array_equal(np.isneginf(v), ref), (hist, seed)
        assert np.isclose(lse, np.log(np.exp(v[np.isfinite(v)].astype(np.float64)).sum()), atol=1e-4)


@pytest.mark.parametrize("case", ["m64", "m128"])
def test_greedy_decode_equals_hf_loop(golden, case):
    """oracle.generate (beam_size=1, T=0: CT2 greedy) vs a cache-free greedy loop built from HF's model + HF's own
    SuppressTokens / SuppressTokensAtBegin / WhisperTimeStamp processors: token-exact, score to 1e-3."""
    _, meta = golden
    g = meta[f"{case}_greedy"]
    spec, w = mg.np_weights(case)
    oracle = omodel.WhisperOracle(omodel.Spec(spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers, spec.dec_layers,
                                              spec.ffn, spec.vocab), w)
    enc = oracle.encode(mg.case_features(case))
    ids = odec.TokenIds(**g["layout"])
    o = odec.GenOptions(ids=ids, beam_size=1, num_hypotheses=1, sampling_temperature=0.0, suppress_blank=True,
                        suppress_tokens=g["suppress"], max_length=len(g["prompt"]) + 24, length_penalty=1.0)
    res = odec.generate(NetProvider(oracle, enc), g["prompt"], o)
    assert res.sequences_ids[0] == g["tokens"]
    assert abs(res.scores[0] * len(g["tokens"]) - g["sum_logprob"]) <= 1e-3 * max(1.0, abs(g["sum_logprob"]))
    assert 0.0 <= res.no_speech_prob <= 1.0


def test_beam_search_contract_on_injected_logits():
    """Properties of the CT2 contract the reference relies on (transcriber_faster_whisper.py:1409-1414):
    beam 1 == greedy; a wider beam never scores worse than greedy under length_penalty=0... and the returned score
    is sum(logp incl. EOT) / len^length_penalty."""
    V = 1711
    L = mg.token_layout(V)
    ids = odec.TokenIds(**L)
    rng = np.random.default_rng(5)
    lg = (rng.standard_normal((30, 5, V)) * 3).astype(np.float32)
    lg[:, :, L["eot"]] += 4.0
    prompt = [L["sot"], L["no_timestamps"]]
    greedy = odec.generate(odec.InjectedLogits(lg[:, :1]), prompt, odec.GenOptions(ids=ids, beam_size=1, max_length=25))
    # replay greedy by hand
    hist, cum = [], np.float32(0)
    o1 = odec.GenOptions(ids=ids, beam_size=1, max_length=25)
    for step in range(23):
        v, lse, _ = odec.process_logits(lg[step, 0], hist, o1, apply_ts=False)
        tok = int(np.argmax(v))
        cum = np.float32(cum + np.float32(v[tok] - lse))
        if tok == L["eot"]:
            break
        hist.append(tok)
    assert greedy.sequences_ids[0] == hist
    assert abs(greedy.scores[0] - float(cum) / max(len(hist), 1)) < 1e-4
    beam = odec.generate(odec.InjectedLogits(lg), prompt, odec.GenOptions(ids=ids, beam_size=5, max_length=25, num_hypotheses=5))
    assert beam.scores == sorted(beam.scores, reverse=True) and 1 <= len(beam.sequences_ids) <= 5
    assert all(L["eot"] not in s for s in beam.sequences_ids)


def test_sampling_is_seeded_and_valid():
    V = 1711
    L = mg.token_layout(V)
    ids = odec.TokenIds(**L)
    lg = (np.random.default_rng(8).standard_normal((20, 5, V)) * 2).astype(np.float32)
    kw = dict(ids=ids, beam_size=1, num_hypotheses=5, sampling_temperature=0.8, max_length=18, suppress_tokens=[3, 4])
    a = odec.generate(odec.InjectedLogits(lg), [L["sot"]], odec.GenOptions(seed=1, **kw))
    b = odec.generate(odec.InjectedLogits(lg), [L["sot"]], odec.GenOptions(seed=1, **kw))
    c = odec.generate(odec.InjectedLogits(lg), [L["sot"]], odec.GenOptions(seed=2, **kw))
    assert a.sequences_ids == b.sequences_ids and a.sequences_ids != c.sequences_ids
    for s in a.sequences_ids:
        assert s[0] >= L["timestamp_begin"] and s[0] <= L["timestamp_begin"] + 50 and 3 not in s and 4 not in s


def _beam_cases():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "beam_golden.json")) as f:
        return json.load(f)["cases"]


class _BiasedProvider(NetProvider):
    """oracle network + the same seeded per-token logit offset the golden HF run applied (make_beam_golden.py)"""

    def __init__(self, model, enc, bias):
        super().__init__(model, enc)
        self.bias = bias

    def prefill(self, tokens):
        r = super().prefill(tokens)
        return None if r is None else r + self.bias

    def step(self, tokens, parents):
        return super().step(tokens, parents) + self.bias


@pytest.mark.parametrize("case", ["m64", "m128"])
def test_beam_search_equals_hf_beam_search(case):
    """The oracle's beam search (= search.hip's decision procedure) against Hugging Face's own beam search
    (`GenerationMixin.generate(num_beams=N, early_stopping=True)` with HF's Whisper logits processors), 10 seeded cases
    per golden model, beams 5 and 3, with and without hypotheses that finish on <|endoftext|>: HF's returned list IS the oracle's set of finished
    hypotheses ranked by HF's length normaliser (same tokens, same order, sums of log-probs to 2e-3). The one deliberate
    difference — CTranslate2 divides by the length WITHOUT the EOT (what transcriber_faster_whisper.py:1412-1414 relies
    on), HF by the length with it — is checked explicitly: the oracle's own order is the CT2-convention order of the
    same set."""
    from tests.golden import make_beam_golden as mb
    spec, w = mg.np_weights(case)
    oracle = omodel.WhisperOracle(omodel.Spec(spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers, spec.dec_layers,
                                              spec.ffn, spec.vocab), w)
    L = mg.token_layout(spec.vocab)
    cases = [c for c in _beam_cases() if c["model"] == case]
    assert len(cases) == mb.N_SEEDS
    n_eot = 0
    for c in cases:
        feats, bias, max_new, beams = mb.case_inputs(case, c["seed"])
        assert (max_new, beams) == (c["max_new_tokens"], c["num_beams"])
        enc = oracle.encode(feats)
        # ask the oracle for EVERY finished hypothesis (a step can finish several at once, so there may be more than
        # `beams` of them); CTranslate2 and HF then keep the best `beams` of that set under their own length normaliser
        o = odec.GenOptions(ids=odec.TokenIds(**L), beam_size=beams, patience=1.0, num_hypotheses=4 * beams, length_penalty=1.0,
                            suppress_blank=True, suppress_tokens=mb.suppress_ids(L), max_length=1 + max_new)
        res = odec.generate(_BiasedProvider(oracle, enc, bias), [L["sot"]], o)
        hf = {tuple(h["tokens"]): h for h in c["hypotheses"]}
        full = {tuple(t): s * max(len(t), 1) for t, s in zip(res.sequences_ids, res.scores)}       # tokens -> sum of log-probs
        assert len(full) == len(res.sequences_ids) >= beams
        ended = {t: len(t) < max_new for t in full}          # shorter than the budget <=> it ended on <|endoftext|>
        # (1) HF's returned list = the best `beams` of the oracle's finished set under HF's normaliser, same order
        by_hf = sorted(full, key=lambda t: -full[t] / (len(t) + (1 if ended[t] else 0)))[:beams]
        assert by_hf == [tuple(h["tokens"]) for h in c["hypotheses"]], (case, c["seed"], by_hf, list(hf))
        for t in by_hf:
            assert ended[t] == hf[t]["ended_with_eot"]
            assert abs(full[t] - hf[t]["sum_logprob"]) <= 2e-3 * max(1.0, abs(hf[t]["sum_logprob"])), (case, c["seed"], t)
            n_eot += hf[t]["ended_with_eot"]
        # (2) the oracle's own order is the CTranslate2 convention: sum / len WITHOUT the EOT (min 1)
        assert [tuple(t) for t in res.sequences_ids] == sorted(full, key=lambda t: -full[t] / max(len(t), 1)), (case, c["seed"])
        # (3) with num_hypotheses = 1 (what the reference asks for) the winner is the head of that order
        o1 = odec.GenOptions(ids=odec.TokenIds(**L), beam_size=beams, patience=1.0, num_hypotheses=1, length_penalty=1.0,
                             suppress_blank=True, suppress_tokens=mb.suppress_ids(L), max_length=1 + max_new)
        best = odec.generate(_BiasedProvider(oracle, enc, bias), [L["sot"]], o1)
        assert best.sequences_ids[0] == res.sequences_ids[0] and abs(best.scores[0] - res.scores[0]) < 1e-6
    assert n_eot >= 5                                  # the finished-hypothesis path was exercised


# ---------------------------------------------------------------- vectors computed BY THE REFERENCE'S OWN log-mel (round 6)
def test_oracle_logmel_against_reference_vectors():
    """tests/golden/ref_logmel_golden.npz holds outputs of whisper_live/transcriber/tensorrt_utils.py::log_mel_spectrogram
    (padding=160) on seeded PCM, generated by tests/golden/make_ref_logmel_golden.py in the build container; tolerance and its
    reason: tests/test_reference_logmel_diff.py (the live differential, which needs /root/reference)."""
    import os
    from whisperlive_amd.synthetic import speech_like_pcm
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_logmel_golden.npz"))
    for i, (sec, seed, n_mels) in enumerate(g["cases"]):
        pcm = speech_like_pcm(float(sec), seed=int(seed))
        want = g[f"logmel_{i}"]
        got = olm.log_mel_spectrogram(pcm, int(n_mels))
        assert got.shape == want.shape
        d = np.abs(got - want)
        assert float(d.max()) <= 1e-4 and float(np.quantile(d, 0.999)) <= 2e-5, (i, float(d.max()))
