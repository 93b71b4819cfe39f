"""GPU parity (-m gpu) of the LARGE-M encoder GEMM (gemm.hip gemm3_kernel: 256 x 256 x 64 tiles, 8 waves, four phases per K tile,
chosen for M >= 4000 rows, i.e. a batched encoder of >= 3 windows): the batched encode of B windows against

  * the single-window encodes of the same clips on the M = 1500 kernel (gemm2_kernel) — both accumulate K in the same order in
    fp32, so the encoder states must agree to the last bit or nearly (<= 1e-5 relative), for EVERY item (256-row tiles straddle
    the item boundaries: 1500 is not a multiple of 256, and the last tile is ragged);
  * the CPU oracle on the same fp16-rounded weights (rel-rms <= 2e-3, the full-depth bound);
  * through the tile-packed cross-attention K / V the same GEMM writes (GEMM_CROSS_KV epilogue): a batched beam-5 decode equals
    the single decodes.
Whisper-small widths (d_model 768: N = 768 / 2304 / 3072, K = 768 / 3072) and large-v3 widths (1280 / 3840 / 5120) at reduced depth."""
import numpy as np
import pytest

from tests import helpers as H
from oracle import logmel as olm
from oracle import model as omodel

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

SHAPES = {
    # name: (n_mels, d_model, heads, enc_layers, dec_layers, ffn, vocab, windows)
    "small-like x5": (80, 768, 12, 2, 2, 3072, 20000, 5),       # M = 7500: 29.3 row tiles
    "large-like x4": (128, 1280, 20, 1, 1, 5120, 20000, 4),     # M = 6000
}


@pytest.mark.parametrize("name", list(SHAPES))
def test_batched_encoder_equals_single_window_encodes_and_oracle(gpu, name):
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import WhisperSpec
    from whisperlive_amd.weights import random_weights
    n_mels, d, h, le, ld, f, v, B = SHAPES[name]
    spec = WhisperSpec(n_mels=n_mels, d_model=d, n_heads=h, enc_layers=le, dec_layers=ld, ffn=f, vocab=v)
    w = random_weights(spec, seed=5)
    eng = HipWhisperEngine(spec, w)
    oracle = omodel.WhisperOracle(H.oracle_spec(spec), H.f16_weights(w))
    sb, s1 = eng.create_slot(B, 5), eng.create_slot(1, 5)
    try:
        clips = [olm.speech_like_pcm(30.0 - 3.0 * i, seed=300 + i) for i in range(B)]
        Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
        segs = [min(t - 1, 3000) for t in Ts]
        sb.encode(B, seek=[0] * B, seg=segs)
        names = None
        ids = H.token_ids_for(spec.vocab)
        kw = dict(beam_size=5, patience=1.0, max_length=1 + 12, suppress_tokens=H.default_suppress(ids))
        res = sb.generate([[ids.sot]] * B, H.engine_ids(ids), **kw)
        worst = 0.0
        for i in range(B):
            T1 = s1.logmel(clips[i])
            assert T1 == Ts[i]
            s1.encode(1, seek=[0], seg=[segs[i]])
            a, b = sb.encoder_output(i), s1.encoder_output(0)
            assert np.isfinite(a).all()
            rel = float(np.abs(a - b).max() / (np.sqrt((b.astype(np.float64) ** 2).mean()) + 1e-30))
            worst = max(worst, rel)
            assert rel <= 1e-5, (name, "item", i, "batched (large-M GEMM) vs single-window encode", rel)
            one = s1.generate([[ids.sot]], H.engine_ids(ids), **kw)[0]
            assert one.sequences_ids == res[i].sequences_ids, (name, "item", i, "decode over the batched cross K/V")
        feats = sb.features(B - 1)
        ref = oracle.encode(olm.pad_or_trim(feats[:, : Ts[B - 1] - 1])[None])[0].numpy()
        st = H.err_stats(sb.encoder_output(B - 1), ref)
        print(name, "batched vs single max rel", worst, "| last item vs oracle", st)
        assert st["rel_rms"] <= 2e-3, st
    finally:
        sb.close(); s1.close(); eng.close()


def test_batched_encoder_is_deterministic_over_repeated_runs(gpu):
    """The persistent large-M GEMM re-stages ring slots while clamped tail requests of the finished tile may still be landing, and
    overlaps the next tile's first K tiles with the epilogue: a race there would show as rare wrong tiles that come and go. Twelve
    batched encodes of the same eight windows must be bit-identical (encoder states of every item), and equal to the single-window encode."""
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.specs import WhisperSpec
    from whisperlive_amd.weights import random_weights
    spec = WhisperSpec(n_mels=80, d_model=768, n_heads=12, enc_layers=2, dec_layers=1, ffn=3072, vocab=20000)
    eng = HipWhisperEngine(spec, random_weights(spec, seed=9))
    B = 8
    sb, s1 = eng.create_slot(B, 5), eng.create_slot(1, 5)
    try:
        clips = [olm.speech_like_pcm(30.0 - 1.7 * i, seed=700 + i) for i in range(B)]
        Ts = [sb.logmel(c, item=i) for i, c in enumerate(clips)]
        segs = [min(t - 1, 3000) for t in Ts]
        first = None
        for rep in range(12):
            sb.encode(B, seek=[0] * B, seg=segs)
            out = np.stack([sb.encoder_output(i) for i in range(B)])
            if first is None:
                first = out
            else:
                assert np.array_equal(out, first), ("run", rep, "differs from run 0 in", int((out != first).sum()), "values")
        for i in (0, 3, B - 1):
            s1.logmel(clips[i])
            s1.encode(1, seek=[0], seg=[segs[i]])
            assert np.array_equal(s1.encoder_output(0), first[i]), ("item", i, "batched vs single-window")
    finally:
        sb.close(); s1.close(); eng.close()
