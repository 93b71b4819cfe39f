"""GPU parity (-m gpu) of the BATCHED decode at full depth (VERDICT r2 'weak' #2): Whisper-large-v3 32 + 32 layers,
8 clips x 5 beams = 40 rows per decode step — the step BASELINE configs[4] (batch_inference.py, 64 clips over 8 GPUs, 8 per
decode: whisper_live/batch_inference.py:225-438) spends its time in — on PEAKED seeded weights (tests/helpers.py), against the
CPU oracle on the same fp16-rounded weights.

The 17..48-row kernels exist only for this mode (K-split slabs at 3 row tiles, `dec_xattn_combine_kernel`, 2-tile fc1, the
raised dynamic-LDS limit at d_model 1280, per-item cross-attention groups of R = 5 rows); the reduced-depth family tests
(tests/test_gpu_lean_family.py) run them on 2 layers. Here:

* the 40 logits rows of a captured batched step, after 5 beam reorders, against the oracle's teacher-forced logits of the
  SAME token histories (rel-rms <= 5e-3): a statement about the kernels that does not depend on how a near-tie was broken;
* 8-item batched beam-5 == the 8 single decodes on the GPU (token-exact, scores 2e-3), and == the oracle for the
  well-conditioned items (token-exact; the GPU's reported score == the oracle's to 5e-3);
* one clip, beam 5, 64 steps, token-exact against the oracle.

The oracle's DECODER is given the GPU's own encoder output for the batched items (8 oracle encoder passes of large-v3 would
take minutes); item 0's encoder output is checked against the oracle's encoder first (rel-rms <= 2e-3), which ties the chain."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from oracle import decoding as odec
from oracle import logmel as olm
from oracle import model as omodel

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1200)]

LOGIT_REL_RMS = 5e-3
NOISE_AMP = 0.02
PEAKED_SEED = 4                      # scripts/scan_peaked_seeds.py large-v3: clip 1, [sot], 64 steps is well-conditioned at +-0.02
PINNED_WELL_CONDITIONED = True
N_ITEMS = 8


@pytest.fixture(scope="module")
def lv3(gpu):
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import SPECS
    spec = SPECS["large-v3"]
    w = H.peaked_weights(spec, PEAKED_SEED)
    eng = HipWhisperEngine(spec, w)
    oracle = omodel.WhisperOracle(H.oracle_spec(spec), H.f16_weights(w))
    del w
    sb = eng.create_slot(N_ITEMS, 5)
    clips = [olm.speech_like_pcm(30.0 - 2.5 * i, seed=900 + i) for i in range(N_ITEMS)]
    Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
    sb.encode(N_ITEMS, seek=[0] * N_ITEMS, seg=[min(t - 1, 3000) for t in Ts])
    encs = [torch.from_numpy(sb.encoder_output(i))[None] for i in range(N_ITEMS)]
    feats0 = sb.features(0)
    ref0 = oracle.encode(olm.pad_or_trim(feats0[:, : Ts[0] - 1])[None])
    st = H.err_stats(encs[0][0].numpy(), ref0[0].numpy())
    print("large-v3 batched encoder item 0 of 8 vs oracle", st)
    assert st["rel_rms"] <= 2e-3, st
    yield spec, eng, oracle, sb, clips, encs
    sb.close()
    eng.close()


def _prefix(a, b):
    n = 0
    while n < min(len(a), len(b)) and a[n] == b[n]:
        n += 1
    return n


def test_40_row_step_logits_after_beam_reorders(lv3):
    spec, eng, oracle, sb, clips, encs = lv3
    ids = H.token_ids_for(spec.vocab)
    K = 6
    kw = dict(beam_size=5, patience=1.0, num_hypotheses=5, max_length=1 + K, suppress_tokens=sorted(H.default_suppress(ids) + [ids.eot]))
    res = sb.generate([[ids.sot]] * N_ITEMS, H.engine_ids(ids), **kw)
    lg = sb.debug_logits(N_ITEMS * 5)                      # the K-th (last) step's logits, rows = item * 5 + beam
    assert np.isfinite(lg).all()
    worst = 0.0
    for i in range(N_ITEMS):
        assert len(res[i].sequences_ids) == 5 and all(len(s) == K for s in res[i].sequences_ids), res[i]
        seqs = np.asarray([[ids.sot] + s[:-1] for s in res[i].sequences_ids])       # what the rows were fed, positions 0..K-1
        ref = oracle.decode_logits(encs[i], seqs)[:, -1].numpy()                    # [5, V] teacher-forced, last position
        rows = lg[5 * i: 5 * i + 5]
        for j in range(5):
            errs = [H.err_stats(rows[r], ref[j])["rel_rms"] for r in range(5)]
            r = int(np.argmin(errs))
            worst = max(worst, errs[r])
            assert errs[r] <= LOGIT_REL_RMS, ("item", i, "hypothesis", j, "best-matching GPU row", r, errs)
            # the token the search appended for this hypothesis must be allowed and plausible under the oracle's row
            assert res[i].sequences_ids[j][-1] != ids.eot
    print("large-v3 40-row step logits vs oracle, worst rel-rms over 40 hypotheses", worst)


def test_eight_items_batched_beam5_equal_singles_and_oracle(lv3):
    spec, eng, oracle, sb, clips, encs = lv3
    ids = H.token_ids_for(spec.vocab)
    STEPS = 24
    kw = dict(beam_size=5, patience=1.0, max_length=1 + STEPS, suppress_tokens=H.default_suppress(ids))
    res = sb.generate([[ids.sot]] * N_ITEMS, H.engine_ids(ids), **kw)
    # singles on the GPU: the same encoder rows (enc_items), one item per decode = the 5-row kernels
    opts = odec.GenOptions(ids=ids, **kw)
    near_ties = 0
    for i in range(N_ITEMS):
        one = sb.generate([[ids.sot]], H.engine_ids(ids), enc_items=[i], **kw)[0]
        if one.sequences_ids != res[i].sequences_ids:
            # Round 6 (log G5): one stream's step sums K = 1280 in four slices of ten k-tiles, the row tiles of a batched step in eight of five —
            # another association of the same sum. A difference between the two GPU decodes is excused only by a near-tie that is SHOWN: the
            # ORACLE's cumulative log-probabilities of the two token sequences (teacher-forced, decoding rules applied) lie within 2 x NOISE_AMP,
            # the noise criterion's own definition of a near-tie (tests/helpers.py check_decode) — and at most one of the eight items may need it.
            a, b = one.sequences_ids[0], res[i].sequences_ids[0]
            la = H.oracle_sequence_logprob(oracle, encs[i], ids, [ids.sot], a, opts, f"single item {i}")
            lb = H.oracle_sequence_logprob(oracle, encs[i], ids, [ids.sot], b, opts, f"batched item {i}")
            print("large-v3 batched item", i, "differs from its single decode after", _prefix(a, b), "tokens; oracle log-probabilities", la, lb)
            assert len(one.sequences_ids) == len(res[i].sequences_ids) and abs(la - lb) <= 2 * NOISE_AMP, ("batched != single", i, _prefix(a, b), la, lb)
            near_ties += 1
        # (scores: the 40-row step and the 5-row step reduce the MLP output projection differently since round 4 — one launch for
        # batched rows, K-split partial sums for a single stream — so the summed log-probabilities differ by fp16-operand
        # rounding: measured 1.05e-3 on the length-normalised score, against 5e-3 allowed versus the oracle)
        assert abs(one.scores[0] - res[i].scores[0]) <= 2e-3
    assert near_ties <= 1, near_ties
    exact = 0
    for i in (0, 3, 7):
        ref = odec.generate(H.NetProvider(oracle, encs[i]), [ids.sot], opts)
        g, r = res[i].sequences_ids[0], ref.sequences_ids[0]
        print("large-v3 batched item", i, "common prefix", _prefix(g, r), "of", len(r), "gpu score", res[i].scores[0], "oracle", ref.scores[0])
        if g == r:
            assert abs(res[i].scores[0] - ref.scores[0]) <= 5e-3
            exact += 1
        else:
            # only a near-tie on the oracle's own decision path excuses a difference (the noise test runs only now)
            assert not H.decode_is_well_conditioned(oracle, encs[i], [ids.sot], opts, ref, NOISE_AMP, seeds=(1,)), (i, _prefix(g, r), g, r)
            lgt = oracle.decode_logits(encs[i], np.asarray([ids.sot] + list(g))[None])[0].numpy()
            cum = 0.0
            for k, t in enumerate(g):
                v, lse, _ = odec.process_logits(lgt[k], list(g[:k]), opts, True)
                assert np.isfinite(v[t])
                cum += float(v[t] - lse)
            assert abs(res[i].scores[0] - cum / max(len(g), 1)) <= 5e-3
            assert cum >= ref.scores[0] * max(len(r), 1) - 5e-2
        assert abs(res[i].no_speech_prob - ref.no_speech_prob) <= 2e-3 + 0.02 * ref.no_speech_prob
    assert exact >= 2, "at least two of the three oracle-checked items must be token-exact"


def test_single_clip_beam5_64_steps_token_exact(lv3):
    spec, eng, oracle, sb, clips, encs = lv3
    ids = H.token_ids_for(spec.vocab)
    H.check_decode(oracle, encs[1], sb, ids, [ids.sot], "large-v3 peaked clip 1, 64 steps", require_exact=PINNED_WELL_CONDITIONED,
                   noise_amp=NOISE_AMP, enc_items=[1], noise_seeds=(1,), beam_size=5, patience=1.0, max_length=1 + 64,
                   suppress_tokens=H.default_suppress(ids))


def test_teacher_forced_40_rows_full_depth(lv3):
    """40 positions of one sequence in ONE pass (three 16-row tiles, R = 16 cross-attention groups) at 32 layers"""
    spec, eng, oracle, sb, clips, encs = lv3
    toks = np.random.default_rng(40).integers(0, spec.vocab, size=40)
    got = sb.debug_decode_logits(toks)
    ref = oracle.decode_logits(encs[0], toks[None])[0].numpy()
    st = H.err_stats(got, ref)
    print("large-v3 peaked teacher-forced 40 rows", st)
    assert st["rel_rms"] <= LOGIT_REL_RMS and st["max_abs"] <= 4 * LOGIT_REL_RMS * st["ref_rms"] + 1e-2, st
