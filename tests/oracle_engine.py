"""An engine-shaped wrapper over the CPU oracle, so the SAME host logic (WhisperModelHIP.transcribe, batch worker) can
be run once on libwlx.so and once on the oracle and the results compared end to end. Test infrastructure only."""
from __future__ import annotations

import threading
from typing import List

import numpy as np

from oracle import decoding as odec
from oracle import logmel as olm
from oracle import model as omodel
from oracle.provider import NetProvider
from whisperlive_amd.engine import GenerationResult


class OracleSlot:
    def __init__(self, engine, max_batch, rows):
        self.engine, self.max_batch, self.rows, self.sid = engine, max_batch, rows, 0
        self.lock = threading.Lock()
        self.feats = {}
        self.enc = None

    def close(self):
        self.sid = -1

    def logmel(self, pcm, item=0):
        self.feats[item] = olm.log_mel_spectrogram(pcm, self.engine.spec.n_mels)
        return self.feats[item].shape[1]

    def set_features(self, feats, item=0):
        self.feats[item] = np.asarray(feats, np.float32)

    def features(self, item=0):
        return self.feats[item]

    def encode(self, batch=1, seek=None, seg=None):
        wins = [olm.pad_or_trim(self.feats[b][:, seek[b]: seek[b] + min(seg[b], 3000)]) for b in range(batch)]
        self.enc = self.engine.oracle.encode(np.stack(wins))

    def encoder_output(self, item=0):
        return self.enc[item].numpy()

    def generate(self, prompts, ids, enc_items=None, **kw) -> List[GenerationResult]:
        o = odec.GenOptions(ids=odec.TokenIds(ids.sot, ids.eot, ids.no_timestamps, ids.timestamp_begin, ids.no_speech, ids.blank), **kw)
        out = []
        for b, p in enumerate(prompts):
            it = enc_items[b] if enc_items is not None else b
            R = max(1, o.num_hypotheses) if (o.sampling_temperature > 0 or o.beam_size <= 1) else o.beam_size
            r = odec.generate(NetProvider(self.engine.oracle, self.enc[it: it + 1]), list(p), o, row_base=b * R)
            out.append(GenerationResult(r.sequences_ids, r.scores, r.no_speech_prob))
        return out

    def align(self, tokens, n_sot, num_frames, heads, eot, median_filter_width=7, item=0):
        from oracle import alignment as oal
        tokens = list(tokens)
        ti, fi, probs, _m = oal.align(self.engine.oracle, self.enc[item: item + 1], tokens[:n_sot], tokens[n_sot], tokens[n_sot + 1:-1],
                                      eot, num_frames, [tuple(h) for h in heads], median_filter_width)
        return np.asarray(ti), np.asarray(fi), np.asarray(probs, np.float32)

    def detect_language(self, batch, sot, lang_ids):
        lg = self.engine.oracle.decode_logits(self.enc[:batch], np.full((batch, 1), sot))[:, 0].numpy()[:, lang_ids]
        e = np.exp(lg - lg.max(axis=1, keepdims=True))
        return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


class OracleEngine:
    def __init__(self, spec, weights_f16_rounded, int8=None):
        self.spec, self.device = spec, 0
        self.oracle = omodel.WhisperOracle(omodel.Spec(spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers,
                                                       spec.dec_layers, spec.ffn, spec.vocab), weights_f16_rounded, int8=int8)

    def create_slot(self, max_batch=1, rows=5):
        return OracleSlot(self, max_batch, rows)
