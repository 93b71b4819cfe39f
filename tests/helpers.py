"""Shared helpers for the parity tests: specs, token-id layouts, oracle<->engine glue."""
from __future__ import annotations

import numpy as np

from oracle import decoding as odec
from oracle import model as omodel
from whisperlive_amd.specs import WhisperSpec, SPECS


# a small architecture that still exercises every code path (2 heads x 64, 2+2 layers, ragged vocab tail)
MICRO = WhisperSpec(n_mels=80, d_model=128, n_heads=2, enc_layers=2, dec_layers=2, ffn=512, vocab=2310)
TINY_EN = SPECS["tiny.en"]


def token_ids_for(vocab: int) -> odec.TokenIds:
    """Whisper id layout scaled to `vocab`: ... eot, sot, (langs), translate, transcribe, sot_lm, sot_prev,
    nospeech, notimestamps, ts_begin .. ts_begin+1500 (= vocab-1). For 51864 this is the real .en layout."""
    ts_begin = vocab - 1501
    return odec.TokenIds(sot=ts_begin - 106, eot=ts_begin - 107, no_timestamps=ts_begin - 1, timestamp_begin=ts_begin,
                         no_speech=ts_begin - 2, blank=220 if vocab > 1000 else 7)


def default_suppress(ids: odec.TokenIds):
    """transcribe/translate/sot/sot_prev/sot_lm-like specials + a few 'non-speech' ids (shape of
    get_suppressed_tokens, transcriber_faster_whisper.py:1831-1853)."""
    tb = ids.timestamp_begin
    return sorted({1, 2, 7, 8, 9, 10, 14, 25, tb - 5, tb - 6, ids.sot, tb - 3, tb - 4})


def oracle_spec(spec: WhisperSpec) -> omodel.Spec:
    return omodel.Spec(spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers, spec.dec_layers, spec.ffn, spec.vocab)


def f16_weights(weights):
    """The engine stores projection matrices in fp16: give the oracle the SAME rounded values so the
    comparison measures arithmetic, not quantisation of the checkpoint."""
    out = {}
    for k, v in weights.items():
        if v.ndim >= 2 and "embed_positions" not in k:
            out[k] = v.astype(np.float16).astype(np.float32)
        else:
            out[k] = v
    return out


from oracle.provider import NetProvider  # noqa: E402,F401


def engine_ids(ids: odec.TokenIds):
    from whisperlive_amd.engine import TokenIds
    return TokenIds(ids.sot, ids.eot, ids.no_timestamps, ids.timestamp_begin, ids.no_speech, ids.blank)


def err_stats(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    return dict(max_abs=float(d.max()), mean_abs=float(d.mean()), ref_rms=float(np.sqrt((b ** 2).mean())),
                rel_rms=float(np.sqrt((d ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30)),
                argmax=int(d.argmax()))
