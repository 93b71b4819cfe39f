"""How far is the HIP fp16 path from the arithmetic the reference's CPU backend really runs?

The reference's CPU default is CTranslate2 `compute_type="int8"` (whisper_live/backend/faster_whisper_backend.py:88-93; REST path
server.py:503-504), and BASELINE.json's north_star asks for transcripts "identical to the faster_whisper CPU backend ... within a stated
WER/logit tolerance". CTranslate2 cannot be installed offline, so the int8 arithmetic is restated in the oracle
(oracle/model.py `WhisperOracle(int8="ct2")`: per-output-row symmetric int8 weights, per-row dynamic int8 activations, integer
accumulation — the scheme CT2 publishes) and the gap is BOUNDED from both sides:

* -m "not gpu": the int8 oracle against the fp32 oracle (logit distance on seeded weights; transcripts of the TRAINED checkpoint,
  tests/golden/trained_tiny — the one place a transcript can be right or wrong);
* -m gpu: on the peaked Whisper-small.en weights (12 + 12 layers, the benchmarked shapes) the HIP path, the int8 oracle and the fp32
  oracle teacher-forced on the SAME 64-token hypothesis: the HIP logits must sit several times closer to fp32 than int8 does, and
  agree with fp32 on the top-1 token at least as often as int8 does.

The numbers these tests print are the stated tolerance in DESIGN.md §2."""
import json
import os

import numpy as np
import pytest

from tests import helpers as H
from oracle import decoding as odec
from oracle import logmel as olm
from oracle import model as omodel

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_tiny")


def test_row_quantiser_is_symmetric_int8_per_output_row():
    import torch
    w = torch.tensor(np.random.default_rng(0).normal(size=(7, 33)).astype(np.float32))
    w[3] *= 50.0                                         # an outlier row must not cost the other rows their resolution
    q, s = omodel.quantize_rows_int8(w)
    assert q.shape == w.shape and s.shape == (7,)
    assert float(q.abs().max()) == 127.0 and bool((q == q.round()).all())
    assert bool((q.abs().amax(dim=1) == 127.0).all())    # every row uses the full int8 range
    back = q / s[:, None]
    assert float(((back - w).abs() / w.abs().amax(dim=1, keepdim=True)).max()) <= 0.5 / 127.0 + 1e-6


def test_int8_oracle_logit_distance_seeded_weights():
    """int8 moves the logits by ~1e-2 relative (fp16 MFMA on the GPU: 6e-4..8e-4 at full depth, tests/test_gpu_long_context.py)"""
    from whisperlive_amd.weights import random_weights
    spec = H.MICRO
    w = H.f16_weights(random_weights(spec, seed=5))
    f32 = omodel.WhisperOracle(H.oracle_spec(spec), w)
    i8 = omodel.WhisperOracle(H.oracle_spec(spec), w, int8="ct2")
    fb = omodel.WhisperOracle(H.oracle_spec(spec), w, int8="fbgemm")
    feats = olm.pad_or_trim(olm.log_mel_spectrogram(olm.speech_like_pcm(4.0, seed=3), spec.n_mels)[:, :-1])[None]
    e32, e8, eb = f32.encode(feats), i8.encode(feats), fb.encode(feats)
    toks = np.random.default_rng(1).integers(0, spec.vocab, size=(1, 12))
    l32 = f32.decode_logits(e32, toks)[0].numpy()
    l8 = i8.decode_logits(e8, toks)[0].numpy()
    lb = fb.decode_logits(eb, toks)[0].numpy()
    se, sl, sb = H.err_stats(e8.numpy(), e32.numpy()), H.err_stats(l8, l32), H.err_stats(lb, l32)
    print("int8 (ct2 scheme) vs fp32: encoder", se["rel_rms"], "logits", sl["rel_rms"], "| torch dynamic qint8 logits", sb["rel_rms"])
    assert 1e-3 <= se["rel_rms"] <= 5e-2 and 1e-3 <= sl["rel_rms"] <= 1e-1, (se, sl)
    assert sb["rel_rms"] <= 2e-1, sb


@pytest.mark.skipif(not os.path.isfile(os.path.join(DIR, "model.safetensors")), reason="tests/golden/trained_tiny not generated")
def test_int8_oracle_transcribes_the_trained_checkpoint():
    """The trained checkpoint through the host logic on the INT8 oracle: the words that were played must come out — word error rate
    within the reference's own bar (WER < 0.05, /root/reference/tests/test_server.py:73-118) of the fp32 / HIP transcripts, which are
    exact (tests/test_trained_tiny.py)."""
    from tests.golden.make_trained_tiny import utterance
    from tests.oracle_engine import OracleEngine
    from tests.test_trained_tiny import KW
    from whisperlive_amd.specs import spec_from_state_dict
    from whisperlive_amd.transcriber import WhisperModelHIP
    from whisperlive_amd.weights import load_model_dir
    sd = load_model_dir(DIR)
    with open(os.path.join(DIR, "expected.json")) as f:
        cases = json.load(f)["cases"][:4]
    model = WhisperModelHIP(DIR, engine=OracleEngine(spec_from_state_dict(sd), H.f16_weights(sd), int8="ct2"))
    n_words = n_err = 0
    for c in cases:
        pcm, ws = utterance(c["seed"])
        segs, _ = model.transcribe(pcm, **KW)
        said = " ".join(s.text for s in segs).split()
        want = [f"w{300 + w}" for w in ws]
        # word-level edit distance
        dp = list(range(len(want) + 1))
        for a in said:
            prev, dp[0] = dp[0], dp[0] + 1
            for j, b in enumerate(want, 1):
                prev, dp[j] = dp[j], min(dp[j] + 1, dp[j - 1] + 1, prev + (a != b))
        n_err += dp[-1]
        n_words += len(want)
    wer = n_err / n_words
    print(f"int8 oracle on the trained checkpoint: word error rate {wer:.3f} ({n_err} of {n_words} words); fp32 oracle / HIP: 0")
    assert wer <= 0.05, wer


@pytest.mark.gpu
def test_hip_fp16_sits_inside_the_int8_gap_small_en(gpu):
    """Whisper-small.en 12 + 12 layers, peaked seeded weights (content-dependent attention, logits with a standard deviation of ~6):
    the fp32 oracle's beam-5 hypothesis of 64 tokens, teacher-forced through (a) the HIP path, (b) the int8 oracle, (c) the fp32
    oracle. Stated tolerance versus the reference's CPU arithmetic: HIP-vs-fp32 logit error <= 1/4 of int8-vs-fp32, and top-1
    agreement with fp32 at least as high as int8's."""
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import SPECS
    spec = SPECS["small.en"]
    w = H.peaked_weights(spec, 42)
    w16 = H.f16_weights(w)
    eng = HipWhisperEngine(spec, w)
    slot = eng.create_slot(1, 5)
    try:
        f32 = omodel.WhisperOracle(H.oracle_spec(spec), w16)
        i8 = omodel.WhisperOracle(H.oracle_spec(spec), w16, int8="ct2")
        pcm = olm.speech_like_pcm(30.0, seed=1234)
        T = slot.logmel(pcm)
        feats = slot.features()
        slot.encode(1, seek=[0], seg=[min(T - 1, 3000)])
        win = olm.pad_or_trim(feats[:, : T - 1])[None]
        e32, e8 = f32.encode(win), i8.encode(win)
        se_hip, se_i8 = H.err_stats(slot.encoder_output(0), e32[0].numpy()), H.err_stats(e8[0].numpy(), e32[0].numpy())
        ids = H.token_ids_for(spec.vocab)
        kw = dict(beam_size=5, patience=1.0, max_length=1 + 64, suppress_tokens=H.default_suppress(ids) + [ids.eot])
        ref = odec.generate(H.NetProvider(f32, e32), [ids.sot], odec.GenOptions(ids=ids, **kw))
        seq = np.asarray([ids.sot] + list(ref.sequences_ids[0]))[:64]
        l32 = f32.decode_logits(e32, seq[None])[0].numpy()
        l8 = i8.decode_logits(e8, seq[None])[0].numpy()
        lhip = slot.debug_decode_logits(seq)
        s_hip, s_i8 = H.err_stats(lhip, l32), H.err_stats(l8, l32)
        top32, top8, toph = l32.argmax(1), l8.argmax(1), lhip.argmax(1)
        agree_hip, agree_i8 = float((toph == top32).mean()), float((top8 == top32).mean())
        # the decodes themselves: common prefix with the fp32 hypothesis
        got = slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0].sequences_ids[0]
        r8 = odec.generate(H.NetProvider(i8, e8), [ids.sot], odec.GenOptions(ids=ids, **kw)).sequences_ids[0]
        pre = lambda a, b: next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
        print(f"small.en peaked: encoder rel-rms HIP {se_hip['rel_rms']:.2e} int8 {se_i8['rel_rms']:.2e} | teacher-forced logits rel-rms "
              f"HIP {s_hip['rel_rms']:.2e} int8 {s_i8['rel_rms']:.2e} (ratio {s_i8['rel_rms'] / s_hip['rel_rms']:.1f}x) | top-1 agreement with fp32 over "
              f"{len(seq)} positions HIP {agree_hip:.3f} int8 {agree_i8:.3f} | beam-5 common prefix with fp32: HIP {pre(got, ref.sequences_ids[0])} "
              f"int8 {pre(r8, ref.sequences_ids[0])} of {len(ref.sequences_ids[0])}")
        assert s_hip["rel_rms"] <= 5e-3 and s_hip["rel_rms"] <= 0.25 * s_i8["rel_rms"], (s_hip, s_i8)
        assert se_hip["rel_rms"] <= 0.25 * se_i8["rel_rms"], (se_hip, se_i8)
        assert agree_hip >= agree_i8 and agree_hip >= 0.98, (agree_hip, agree_i8)
        assert pre(got, ref.sequences_ids[0]) >= pre(r8, ref.sequences_ids[0])
    finally:
        slot.close()
        eng.close()
