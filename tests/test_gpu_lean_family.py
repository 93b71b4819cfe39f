"""GPU parity (-m gpu) of the LEAN decode kernels (decoder.hip dec_gemv2_kernel and friends) at the d_model values of
the Whisper family — 384 (tiny, since round 5), 512 (base), 768 (small), 1024 (medium), 1280 (large-v3) — with reduced depth so
the CPU oracle finishes in seconds: every size the reference lists (whisper_live/backend/faster_whisper_backend.py:74-79) decodes on
the lean kernels; the first-generation GEMV (tests/test_gpu_parity.py: d_model 128) stays for shapes outside the family. This module
is what pins the kernels bench.py times.

Tolerances as at full depth (tests/test_gpu_full_depth.py): logits rel-rms <= 5e-3 and max-abs <= 2e-2 * rms + 1e-2 against the fp32 oracle on
the same fp16-rounded weights; generated tokens equal up to the first fp16-vs-fp32 near-tie (>= 6 tokens)."""
import numpy as np
import pytest

from tests import helpers as H
from oracle import decoding as odec
from oracle import logmel as olm
from oracle import model as omodel

pytestmark = pytest.mark.gpu

FAMILY = {
    # name: (n_mels, d_model, heads, enc_layers, dec_layers, ffn, vocab)
    "tiny-like": (80, 384, 6, 1, 2, 1536, 20000),       # round 5: d_model 384 = 1.5 x 256 on the lean kernels too (LNV = 15, six waves of two k-tiles)
    "base-like": (80, 512, 8, 1, 2, 2048, 20000),
    "small-like": (80, 768, 12, 1, 2, 3072, 51864),     # full vocabulary: the 2-tile vocabulary projection
    "small-multilingual-like": (80, 768, 12, 1, 2, 3072, 51865),    # BASELINE configs[2]: the multilingual vocabulary is NOT a multiple of 16 (ragged last tile of the vocabulary projection, of the search's last timestamp chunk)
    "medium-like": (80, 1024, 16, 1, 2, 4096, 20000),
    "large-like": (128, 1280, 20, 1, 2, 5120, 20000),
}


@pytest.fixture(scope="module", params=list(FAMILY))
def fam(request, gpu):
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import WhisperSpec
    from whisperlive_amd.weights import random_weights
    n_mels, d, h, le, ld, f, v = FAMILY[request.param]
    spec = WhisperSpec(n_mels=n_mels, d_model=d, n_heads=h, enc_layers=le, dec_layers=ld, ffn=f, vocab=v)
    w = random_weights(spec, seed=11)
    eng = HipWhisperEngine(spec, w)
    oracle = omodel.WhisperOracle(H.oracle_spec(spec), H.f16_weights(w))
    slot = eng.create_slot(1, 5)
    pcm = olm.speech_like_pcm(5.0, seed=21)
    T = slot.logmel(pcm)
    feats = slot.features()
    slot.encode(1, seek=[0], seg=[T - 1])
    enc = oracle.encode(olm.pad_or_trim(feats[:, : T - 1])[None])
    yield request.param, spec, eng, oracle, slot, enc
    slot.close()
    eng.close()


def _check(got, ref, what):
    st = H.err_stats(got, ref)
    assert np.isfinite(np.asarray(got)).all(), what
    assert st["rel_rms"] <= 5e-3 and st["max_abs"] <= 2e-2 * st["ref_rms"] + 1e-2, (what, st)
    return st


def test_encoder_parity(fam):
    name, spec, eng, oracle, slot, enc = fam
    _check(slot.encoder_output(0), enc[0].numpy(), f"{name} encoder")


@pytest.mark.parametrize("n_tok", [1, 5, 16, 20, 32, 40, 48])
def test_decoder_logits_rows_1_to_48(fam, n_tok):
    """teacher-forced rows in ONE pass: 1..16 rows = one MFMA row tile, 17..32 = two, 33..48 = three (batched streams;
    at d_model 1280 the staged rows exceed the default 64 KiB of dynamic LDS: the raised-limit path)."""
    name, spec, eng, oracle, slot, enc = fam
    toks = np.random.default_rng(n_tok).integers(0, spec.vocab, size=n_tok)
    got = slot.debug_decode_logits(toks)
    ref = oracle.decode_logits(enc, toks[None])[0].numpy()
    print(name, n_tok, _check(got, ref, f"{name} logits n={n_tok}"))


def test_beam_decode_steps(fam):
    """the captured decode-step graph (5 beam rows): lean GEMVs, self/cross attention, device-side beam search."""
    name, spec, eng, oracle, slot, enc = fam
    names = [k["name"] for k in slot.debug_profile_step(5, 4, 2)]
    assert any(n.startswith("dec_gemv2_kernel<") for n in names) and any(n.startswith("dec_vocab_kernel<") for n in names), names
    assert not any(n.startswith("dec_gemv_kernel<") for n in names), (name, names)     # no family member on the first-generation kernel
    ids = H.token_ids_for(spec.vocab)
    kw = dict(beam_size=5, patience=1.0, max_length=1 + 16, suppress_tokens=H.default_suppress(ids))
    got = slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0]
    again = slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0]
    assert got.sequences_ids == again.sequences_ids and got.scores == again.scores      # deterministic
    ref = odec.generate(H.NetProvider(oracle, enc), [ids.sot], odec.GenOptions(ids=ids, **kw))
    g, r = got.sequences_ids[0], ref.sequences_ids[0]
    n = 0
    while n < min(len(g), len(r)) and g[n] == r[n]:
        n += 1
    assert n >= min(6, len(r)), (name, g, r)
    assert abs(got.no_speech_prob - ref.no_speech_prob) <= 5e-3 + 0.05 * ref.no_speech_prob


def test_five_items_batched_equal_singles(fam):
    """25 beam rows in one decode (two row tiles) == each clip decoded alone."""
    name, spec, eng, oracle, slot, enc = fam
    ids = H.token_ids_for(spec.vocab)
    kw = dict(beam_size=5, max_length=1 + 10, suppress_tokens=H.default_suppress(ids))
    clips = [olm.speech_like_pcm(3.0 + 0.5 * i, seed=60 + i) for i in range(5)]
    singles = []
    for c in clips:
        T = slot.logmel(c); slot.encode(1, seek=[0], seg=[T - 1])
        singles.append(slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0])
    sb = eng.create_slot(5, 5)
    try:
        Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
        sb.encode(5, seek=[0] * 5, seg=[t - 1 for t in Ts])
        res = sb.generate([[ids.sot]] * 5, H.engine_ids(ids), **kw)
        for i in range(5):
            assert res[i].sequences_ids == singles[i].sequences_ids, (name, i)
            assert abs(res[i].scores[0] - singles[i].scores[0]) < 1e-3
    finally:
        sb.close()


def test_eight_items_batched_equal_singles(fam):
    """40 beam rows in one decode (three row tiles: the batch worker's default of 8 clips) == each clip decoded alone."""
    name, spec, eng, oracle, slot, enc = fam
    ids = H.token_ids_for(spec.vocab)
    kw = dict(beam_size=5, max_length=1 + 8, suppress_tokens=H.default_suppress(ids))
    clips = [olm.speech_like_pcm(2.0 + 0.5 * i, seed=80 + i) for i in range(8)]
    singles = []
    for c in clips:
        T = slot.logmel(c); slot.encode(1, seek=[0], seg=[T - 1])
        singles.append(slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0])
    sb = eng.create_slot(8, 5)
    try:
        Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
        sb.encode(8, seek=[0] * 8, seg=[t - 1 for t in Ts])
        res = sb.generate([[ids.sot]] * 8, H.engine_ids(ids), **kw)
        for i in range(8):
            assert res[i].sequences_ids == singles[i].sequences_ids, (name, i)
            assert abs(res[i].scores[0] - singles[i].scores[0]) < 1e-3
    finally:
        sb.close()


@pytest.mark.parametrize("seed,beam", [(0, 5), (1, 5), (2, 3), (3, 4), (4, 5)])
def test_search_on_injected_logits_full_vocabulary(fam, seed, beam):
    """the device search (chunked scan + merge/update, search.hip) on INJECTED logits at the real vocabulary size (27 chunks
    per row; the micro model of test_gpu_parity.py has two), token-exact against the oracle's beam search: ties included —
    logits are quantised to a coarse grid so that equal values (broken by the smaller id) occur in every list, several of
    the best candidates are planted in ONE lane / one chunk / adjacent chunks, and a few steps have fewer than 2 x beam
    allowed ids in most chunks."""
    name, spec, eng, oracle, slot, enc = fam
    if spec.vocab < 50000:
        pytest.skip("full-vocabulary case: the small-like member of the family")
    ids = H.token_ids_for(spec.vocab)
    rng = np.random.default_rng(100 + seed)
    steps, V = 14, spec.vocab
    lg = (rng.standard_normal((steps, beam, V)) * 3.0).astype(np.float32)
    lg = np.round(lg * 4.0) / 4.0                                   # ties everywhere
    lg[:, :, ids.timestamp_begin:] += 2.0
    lg[:, :, ids.eot] += (5.0 if seed % 2 else 0.0)
    for t in range(steps):                                          # clustered winners
        base = int(rng.integers(1000, V - 9000))
        lg[t, :, base:base + 2048 * 2:256] += 9.0                   # same lane of a chunk (stride = threads per scan workgroup)
        lg[t, :, base + 7:base + 19] += 8.75                        # neighbours inside one chunk
    lg[5:7, :, 3000:ids.eot - 200] = -np.inf                        # nearly empty text chunks
    prompt = [ids.sot]
    kw = dict(beam_size=beam, patience=1.0, max_length=1 + steps - 2, suppress_tokens=H.default_suppress(ids), length_penalty=1.0)
    ref = odec.generate(odec.InjectedLogits(lg), prompt, odec.GenOptions(ids=ids, **kw))
    got = slot.debug_search(lg, prompt, H.engine_ids(ids), **kw)
    assert got.sequences_ids == ref.sequences_ids, (got.sequences_ids, ref.sequences_ids)
    np.testing.assert_allclose(got.scores, ref.scores, rtol=1e-3, atol=1e-3)


def test_twelve_items_batched_equal_singles(fam):
    """60 beam rows in one decode — more than the 48 rows one launch of the lean kernels holds: every projection runs as two
    row chunks (48 + 12) in grid.z (round 3; until then 49..64 rows fell back to the first-generation kernel) — == each clip
    decoded alone; and 60 teacher-forced rows in one pass against the oracle."""
    name, spec, eng, oracle, slot, enc = fam
    ids = H.token_ids_for(spec.vocab)
    kw = dict(beam_size=5, max_length=1 + 8, suppress_tokens=H.default_suppress(ids))
    clips = [olm.speech_like_pcm(2.0 + 0.25 * i, seed=120 + i) for i in range(12)]
    singles = []
    for c in clips:
        T = slot.logmel(c); slot.encode(1, seek=[0], seg=[T - 1])
        singles.append(slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0])
    sb = eng.create_slot(12, 5)
    try:
        Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
        sb.encode(12, seek=[0] * 12, seg=[t - 1 for t in Ts])
        res = sb.generate([[ids.sot]] * 12, H.engine_ids(ids), **kw)
        for i in range(12):
            assert res[i].sequences_ids == singles[i].sequences_ids, (name, i)
            assert abs(res[i].scores[0] - singles[i].scores[0]) < 1e-3
    finally:
        sb.close()
    # restore the fixture's encoder state (item 0 of the shared slot) and check 60 rows in ONE teacher-forced pass
    pcm = olm.speech_like_pcm(5.0, seed=21)
    T = slot.logmel(pcm); slot.encode(1, seek=[0], seg=[T - 1])
    import os
    old = os.environ.get("WLX_PREFILL_ROWS")
    os.environ["WLX_PREFILL_ROWS"] = "64"                 # the debug hook's pass takes up to 64 rows per chunk: 60 rows = one pass
    try:
        toks = np.random.default_rng(60).integers(0, spec.vocab, size=60)
        got = slot.debug_decode_logits(toks)
    finally:
        if old is None:
            os.environ.pop("WLX_PREFILL_ROWS")
        else:
            os.environ["WLX_PREFILL_ROWS"] = old
    ref = oracle.decode_logits(enc, toks[None])[0].numpy()
    print(name, 60, _check(got, ref, f"{name} logits n=60"))


def test_twentyfour_items_batched_equal_singles_and_120_rows(fam):
    """Round 5 — the 64-row cap lifted (the reference's worker takes any max_batch_size, whisper_live/batch_inference.py:113-121): 24 clips x
    5 beams = 120 decoder rows in ONE step — eight 16-row tiles per projection, the vocabulary projection in two row chunks (64 + 56),
    item prompts prefilled in blocks — equal to each clip decoded alone; and 120 teacher-forced rows in one pass against the oracle."""
    name, spec, eng, oracle, slot, enc = fam
    ids = H.token_ids_for(spec.vocab)
    kw = dict(beam_size=5, max_length=1 + 6, suppress_tokens=H.default_suppress(ids))
    clips = [olm.speech_like_pcm(5.0, seed=21)] + [olm.speech_like_pcm(2.0 + 0.125 * i, seed=320 + i) for i in range(1, 24)]
    singles = []
    for c in clips:
        T = slot.logmel(c); slot.encode(1, seek=[0], seg=[T - 1])
        singles.append(slot.generate([[ids.sot]], H.engine_ids(ids), **kw)[0])
    sb = eng.create_slot(24, 5)
    try:
        Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
        sb.encode(24, seek=[0] * 24, seg=[t - 1 for t in Ts])
        res = sb.generate([[ids.sot]] * 24, H.engine_ids(ids), **kw)
        for i in range(24):
            assert res[i].sequences_ids == singles[i].sequences_ids, (name, i)
            assert abs(res[i].scores[0] - singles[i].scores[0]) < 2e-3
        names = [k["name"] for k in sb.debug_profile_step(120, 4, 2)]
        assert not any(n.startswith("dec_gemv_kernel<") for n in names), names          # every projection of the 120-row step on the lean kernels
        # 120 teacher-forced rows of item 0 (= the fixture's clip) in ONE pass through the same row-tiled kernels
        import os
        old = os.environ.get("WLX_PREFILL_ROWS")
        os.environ["WLX_PREFILL_ROWS"] = "120"
        try:
            toks = np.random.default_rng(120).integers(0, spec.vocab, size=120)
            got = sb.debug_decode_logits(toks)
        finally:
            if old is None:
                os.environ.pop("WLX_PREFILL_ROWS")
            else:
                os.environ["WLX_PREFILL_ROWS"] = old
        ref = oracle.decode_logits(enc, toks[None])[0].numpy()
        print(name, 120, _check(got, ref, f"{name} logits n=120"))
    finally:
        sb.close()
    pcm = olm.speech_like_pcm(5.0, seed=21)
    T = slot.logmel(pcm); slot.encode(1, seek=[0], seg=[T - 1])


def test_thirtytwo_items_with_short_prompts_prefill_in_blocks(fam):
    """batch_inference batches of multilingual requests, wider than the prefill working set: every item carries `[sot, lang, task]`
    (+ a prefix); 32 items x 16 prompt rows pass the 448-row prefill buffers, so the joint prefill runs in two blocks of 28 + 4 items
    (engine.hip generate_impl, round 5). Every item must equal its single decode — tokens, score, and no_speech_prob, which is read from the
    prefill logits at the item's <|startoftranscript|> row — also across the block boundary. Then detect_language on the 32-item batch
    (one decoder pass of 32 rows, R = 1) against per-item calls."""
    name, spec, eng, oracle, slot, enc = fam
    if name not in ("base-like", "small-like"):
        pytest.skip("one narrow and one full-vocabulary member are enough for the block loop")
    ids = H.token_ids_for(spec.vocab)
    lang, task = ids.sot + 1, ids.timestamp_begin - 5
    forms = [[ids.sot, lang, task], [ids.sot, lang + 2, task, 1100, 1200], [ids.sot, lang + 1, task, 77]]
    prompts = [forms[i % 3] for i in range(32)]
    clips = [olm.speech_like_pcm(2.0 + 0.1 * i, seed=500 + i) for i in range(32)]
    kw = dict(beam_size=5, patience=1.0, max_length=max(len(p) for p in prompts) + 6, suppress_tokens=H.default_suppress(ids))
    sb = eng.create_slot(32, 5)
    try:
        Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
        sb.encode(32, seek=[0] * 32, seg=[t - 1 for t in Ts])
        res = sb.generate(prompts, H.engine_ids(ids), **kw)
        near_ties = 0
        for i in (0, 1, 2, 13, 26, 27, 28, 29, 31):              # both sides of the 28-item block boundary
            one = sb.generate([prompts[i]], H.engine_ids(ids), enc_items=[i], **kw)[0]
            # the joint pass and the per-item pass are two shapes of the same prompt rows (16-row groups vs one short pass): a few fp16
            # roundings of the cached K / V apart. On these flat seeded weights that can flip a near-tie among the <= 51 first-timestamp
            # candidates; then the two hypotheses must score the same (a WRONG cache row / position moves the score by O(1))
            assert abs(one.scores[0] - res[i].scores[0]) <= 3e-3, (name, i, one.scores[0], res[i].scores[0], one.sequences_ids[0], res[i].sequences_ids[0])
            near_ties += one.sequences_ids[0] != res[i].sequences_ids[0]
            assert abs(one.no_speech_prob - res[i].no_speech_prob) <= 1e-5 + 2e-2 * one.no_speech_prob, (name, i)
        assert near_ties <= 1, (name, near_ties)
        lang_ids = [ids.sot + 1 + j for j in range(8)]
        probs = sb.detect_language(32, ids.sot, lang_ids)
        assert probs.shape == (32, 8) and np.allclose(probs.sum(axis=1), 1.0, atol=1e-4)
        s1 = eng.create_slot(1, 5)
        try:
            for i in (0, 17, 31):
                T = s1.logmel(clips[i]); s1.encode(1, seek=[0], seg=[T - 1])
                np.testing.assert_allclose(s1.detect_language(1, ids.sot, lang_ids)[0], probs[i], rtol=2e-3, atol=2e-5)
        finally:
            s1.close()
    finally:
        sb.close()
    pcm = olm.speech_like_pcm(5.0, seed=21)
    T = slot.logmel(pcm); slot.encode(1, seek=[0], seg=[T - 1])


def test_busy_device_launch_shapes_give_identical_results(fam):
    """With three or more live slots on the device the engine captures a second step graph per slot whose row-tiled residual projections
    take two 16-column tiles per workgroup (work-saving shapes for a work-bound GPU: engine.hip device_is_busy, decoder.hip gemv2_cfg). The
    tile grouping does not touch any summation order: an 8-item batched beam-5 decode must give the SAME tokens and bit-identical scores
    with and without the extra live slots."""
    name, spec, eng, oracle, slot, enc = fam
    ids = H.token_ids_for(spec.vocab)
    kw = dict(beam_size=5, max_length=1 + 8, suppress_tokens=H.default_suppress(ids))
    clips = [olm.speech_like_pcm(2.0 + 0.3 * i, seed=220 + i) for i in range(8)]
    sb = eng.create_slot(8, 5)                 # live slots: the fixture's + this one = 2 -> lone-slot shapes
    extra = []
    try:
        Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
        sb.encode(8, seek=[0] * 8, seg=[t - 1 for t in Ts])
        lone = sb.generate([[ids.sot]] * 8, H.engine_ids(ids), **kw)
        extra = [eng.create_slot(1, 5) for _ in range(2)]          # 4 live slots: the busy-device variant from the next call on
        busy = sb.generate([[ids.sot]] * 8, H.engine_ids(ids), **kw)
        names = [k["name"] for k in sb.debug_profile_step(40, 4, 2)]
        for i in range(8):
            assert busy[i].sequences_ids == lone[i].sequences_ids, (name, i)
            assert busy[i].scores[0] == lone[i].scores[0], (name, i, busy[i].scores[0], lone[i].scores[0])
        # the profile of a 40-row step taken now lists a two-tile fp16-rows-in residual projection (template arguments ..., IN 1, OUT 3, NTB 2, MT 1, XS 0)
        import os
        if os.environ.get("WLX_ROWTILE", "1") != "0" and os.environ.get("WLX_ROWTILE_CHUNK", "16") == "16" and os.environ.get("WLX_RT_F16_NTB2", "") != "0":
            assert any(n.startswith("dec_gemv2_kernel<") and n.endswith(", 1, 3, 2, 1, 0>") for n in names), names
    finally:
        for x in extra:
            x.close()
        sb.close()
    # restore the fixture's encoder state (item 0 of the shared slot)
    pcm = olm.speech_like_pcm(5.0, seed=21)
    T = slot.logmel(pcm); slot.encode(1, seek=[0], seg=[T - 1])
