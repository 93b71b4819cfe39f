"""Oracle network as a logits provider for oracle.decoding.generate. Test infrastructure only — see
oracle/__init__.py. Plays the role of the decoder inside ctranslate2.models.Whisper.generate
(whisper_live/transcriber/transcriber_faster_whisper.py:1394-1407) for the CPU restatement of the search."""
from __future__ import annotations

import numpy as np

from . import decoding as odec
from . import model as omodel


class NetProvider(odec.LogitsProvider):
    def __init__(self, model: omodel.WhisperOracle, enc):
        self.dec = omodel.StepDecoder(model, enc)

    def prefill(self, tokens):
        if len(tokens) == 0:
            return None
        return self.dec.step(np.asarray(tokens)[None, :])[0].numpy()

    def step(self, tokens, parents):
        if self.dec.k[0] is not None and self.dec.k[0].shape[0] == 1 and len(parents) > 1:
            self.dec.reorder([0] * len(parents))
        elif self.dec.k[0] is not None:
            self.dec.reorder(list(parents))
        return self.dec.step(np.asarray(tokens)[:, None])[:, 0].numpy()
