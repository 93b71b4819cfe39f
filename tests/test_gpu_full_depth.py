"""GPU parity (-m gpu) at the BENCHMARKED depth (VERDICT r01 "weak" #1/#2): Whisper-small.en 12+12 layers — the exact
configuration bench.py times — and Whisper-large-v3 32+32, against the CPU oracle on the same fp16-rounded seeded
weights, with tolerances ~10x above what the kernels deliver instead of ~100x:

* encoder states: relative RMS <= 2e-3 (fp16 MFMA operands, fp32 accumulation, fp32 residual stream);
* teacher-forced decoder logits (1 / 5 / 32 rows): relative RMS <= 5e-3, max-abs <= 2e-2 * rms + 1e-2;
* beam-5 decode through the captured decode-step graph, 16 steps: token-exact against the oracle, or — since seeded
  random weights give near-flat distributions — the GPU's sequence must be an equally good hypothesis UNDER THE ORACLE
  (teacher-forced oracle score of the GPU's tokens within 2e-2 of the oracle's own best), and in every case the score
  the GPU reports for its sequence must equal the oracle's evaluation of that same sequence to 5e-3.

What this catches that the reduced-depth family tests cannot: layer-stride arithmetic (kv_layer_stride, the fused
N = L*2*d cross-K/V GEMM, per-layer weight offsets), error growth over 12 / 32 layers, a dropped bias or wrong q-scale
in any one layer."""
import numpy as np
import pytest

from tests import helpers as H
from oracle import decoding as odec
from oracle import logmel as olm
from oracle import model as omodel

pytestmark = pytest.mark.gpu

ENC_REL_RMS = 2e-3
LOGIT_REL_RMS = 5e-3


def _build(name, seed):
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import SPECS
    from whisperlive_amd.weights import random_weights
    spec = SPECS[name]
    w = random_weights(spec, seed=seed)
    eng = HipWhisperEngine(spec, w)
    oracle = omodel.WhisperOracle(H.oracle_spec(spec), H.f16_weights(w))
    del w
    slot = eng.create_slot(1, 5)
    pcm = olm.speech_like_pcm(30.0, seed=1234)          # bench.py's window
    T = slot.logmel(pcm)
    feats = slot.features()
    slot.encode(1, seek=[0], seg=[min(T - 1, 3000)])
    enc = oracle.encode(olm.pad_or_trim(feats[:, : T - 1])[None])
    return spec, eng, oracle, slot, enc


@pytest.fixture(scope="module")
def small_en(gpu):
    spec, eng, oracle, slot, enc = _build("small.en", 0)   # seed 0 = bench.py's weights
    yield spec, eng, oracle, slot, enc
    slot.close()
    eng.close()


def _close(got, ref, rel, what):
    st = H.err_stats(got, ref)
    assert np.isfinite(np.asarray(got)).all(), what
    assert st["rel_rms"] <= rel and st["max_abs"] <= 4 * rel * st["ref_rms"] + 1e-2, (what, st)
    return st


def _oracle_score(oracle, enc, prompt, toks, opts):
    """teacher-forced sum of processed log-probs of `toks` after `prompt` under the oracle network + logits rules"""
    seq = list(prompt) + list(toks)
    lg = oracle.decode_logits(enc, np.asarray(seq)[None])[0].numpy()
    apply_ts = opts.ids.no_timestamps not in prompt
    cum = 0.0
    for i, t in enumerate(toks):
        v, lse, _ = odec.process_logits(lg[len(prompt) - 1 + i], list(toks[:i]), opts, apply_ts)
        assert np.isfinite(v[t]), ("GPU emitted a token the rules forbid", i, t)
        cum += float(v[t] - lse)
    return cum


def _check_beam(oracle, enc, slot, spec, steps, what):
    ids = H.token_ids_for(spec.vocab)
    kw = dict(beam_size=5, patience=1.0, max_length=1 + steps, suppress_tokens=H.default_suppress(ids))
    got = slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0]
    opts = odec.GenOptions(ids=ids, **kw)
    ref = odec.generate(H.NetProvider(oracle, enc), [ids.sot], opts)
    g, r = got.sequences_ids[0], ref.sequences_ids[0]
    n = 0
    while n < min(len(g), len(r)) and g[n] == r[n]:
        n += 1
    sg = _oracle_score(oracle, enc, [ids.sot], g, opts)
    # the score the GPU reports for ITS sequence vs the oracle's evaluation of the same tokens (length_penalty 1)
    assert abs(got.scores[0] - sg / max(len(g), 1)) <= 5e-3, (what, got.scores[0], sg / max(len(g), 1))
    if g != r:
        sr = ref.scores[0] * max(len(r), 1)
        assert len(g) == len(r) and sg >= sr - 2e-2, (what, "GPU sequence is not a near-tie of the oracle's best", n, g, r, sg, sr)
    assert abs(got.no_speech_prob - ref.no_speech_prob) <= 2e-3 + 0.02 * ref.no_speech_prob
    print(what, "beam-5 common prefix", n, "of", len(r), "gpu score", got.scores[0], "oracle score", ref.scores[0])
    return n, len(r)


def test_small_en_full_depth_encoder(small_en):
    spec, eng, oracle, slot, enc = small_en
    print("small.en 12-layer encoder", _close(slot.encoder_output(0), enc[0].numpy(), ENC_REL_RMS, "small.en encoder"))


@pytest.mark.parametrize("n_tok", [1, 5, 32])
def test_small_en_full_depth_logits(small_en, n_tok):
    spec, eng, oracle, slot, enc = small_en
    toks = np.random.default_rng(100 + n_tok).integers(0, spec.vocab, size=n_tok)
    got = slot.debug_decode_logits(toks)
    ref = oracle.decode_logits(enc, toks[None])[0].numpy()
    print("small.en 12-layer decoder rows", n_tok, _close(got, ref, LOGIT_REL_RMS, f"small.en logits n={n_tok}"))


def test_small_en_full_depth_beam5(small_en):
    spec, eng, oracle, slot, enc = small_en
    _check_beam(oracle, enc, slot, spec, 16, "small.en")


def test_small_en_bench_decode_matches_oracle_32_steps(small_en):
    """the benchmark's own decode (EOT suppressed, bench.py's suppress list) for 32 steps — what bench.py reports as
    parity_prefix — held to the same near-tie standard"""
    import bench
    spec, eng, oracle, slot, enc = small_en
    ids_d = bench.token_ids(spec.vocab)
    ids = odec.TokenIds(**ids_d)
    kw = dict(beam_size=5, patience=1.0, max_length=1 + 32, suppress_tokens=bench.suppress_list(ids_d, True))
    got = slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0]
    opts = odec.GenOptions(ids=ids, **kw)
    ref = odec.generate(H.NetProvider(oracle, enc), [ids.sot], opts)
    g, r = got.sequences_ids[0], ref.sequences_ids[0]
    assert len(g) == len(r) == 32
    sg = _oracle_score(oracle, enc, [ids.sot], g, opts)
    assert abs(got.scores[0] - sg / 32) <= 5e-3
    assert g == r or sg >= ref.scores[0] * 32 - 2e-2, (g, r, sg, ref.scores[0] * 32)


def test_engine_from_torch_device_tensors_is_bit_identical(small_en):
    """north_star: 'PyTorch-ROCm holding the weight tensors only' — the engine built from torch.cuda tensors
    (wlx_tensor.on_device = 1: device pointers borrowed for the repack) must produce the same bits as the engine built
    from host numpy arrays."""
    import torch
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.weights import random_weights
    spec, eng, oracle, slot, enc = small_en
    w = random_weights(spec, seed=0)
    tw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    # mixed dtypes on the device side as a checkpoint loader would leave them: fp16 matrices, fp32 vectors
    tw = {k: (t.half() if t.ndim >= 2 and "embed_positions" not in k else t) for k, t in tw.items()}
    eng2 = HipWhisperEngine(spec, tw)
    s2 = eng2.create_slot(1, 5)
    try:
        pcm = olm.speech_like_pcm(30.0, seed=1234)
        T = s2.logmel(pcm)
        s2.encode(1, seek=[0], seg=[min(T - 1, 3000)])
        assert np.array_equal(s2.encoder_output(0), slot.encoder_output(0))
        toks = np.random.default_rng(5).integers(0, spec.vocab, size=5)
        assert np.array_equal(s2.debug_decode_logits(toks), slot.debug_decode_logits(toks))
    finally:
        s2.close()
        eng2.close()


@pytest.mark.timeout(900)
def test_large_v3_full_depth_encoder_and_logits(gpu):
    """32 + 32 layers, 128 mels, d_model 1280, vocab 51866: encoder states + 5-row teacher-forced logits."""
    spec, eng, oracle, slot, enc = _build("large-v3", 1)
    try:
        print("large-v3 32-layer encoder", _close(slot.encoder_output(0), enc[0].numpy(), ENC_REL_RMS, "large-v3 encoder"))
        toks = np.random.default_rng(7).integers(0, spec.vocab, size=5)
        got = slot.debug_decode_logits(toks)
        ref = oracle.decode_logits(enc, toks[None])[0].numpy()
        print("large-v3 32-layer decoder rows 5", _close(got, ref, LOGIT_REL_RMS, "large-v3 logits n=5"))
        _check_beam(oracle, enc, slot, spec, 8, "large-v3")
    finally:
        slot.close()
        eng.close()
