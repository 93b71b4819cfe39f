"""Scripted stand-ins for the engine so the HOST logic (seek loop, prompts, fallback, batching, sessions) can be tested
on CPU. They compute nothing: outputs are scripted per call, like the MagicMock transcriber of the reference's
tests/test_batch_inference.py:52-78."""
from __future__ import annotations

import threading
from typing import Callable, List, Optional

import numpy as np

from whisperlive_amd.engine import GenerationResult
from whisperlive_amd.specs import WhisperSpec


class FakeSlot:
    def __init__(self, engine, max_batch, rows):
        self.engine, self.max_batch, self.rows, self.sid = engine, max_batch, rows, 0
        self.lock = threading.Lock()
        self.calls: List[tuple] = []
        self._frames = {}

    def close(self):
        self.sid = -1

    def logmel(self, pcm, item=0):
        t = (len(pcm) + 160) // 160
        self._frames[item] = t
        self.calls.append(("logmel", item, len(pcm)))
        return t

    def set_features(self, feats, item=0):
        self._frames[item] = feats.shape[-1]
        self.calls.append(("set_features", item, feats.shape))

    def features(self, item=0):
        return np.zeros((self.engine.spec.n_mels, self._frames[item]), np.float32)

    def encode(self, batch=1, seek=None, seg=None):
        self.calls.append(("encode", batch, list(seek), list(seg)))

    def encoder_output(self, item=0):
        return np.zeros((1500, self.engine.spec.d_model), np.float32)

    def generate(self, prompts, ids, **kw):
        self.calls.append(("generate", [list(p) for p in prompts], dict(kw)))
        return self.engine.script_generate(self, prompts, ids, kw)

    def detect_language(self, batch, sot, lang_ids):
        self.calls.append(("detect_language", batch))
        return self.engine.script_lang(batch, lang_ids)


class FakeEngine:
    def __init__(self, spec: Optional[WhisperSpec] = None):
        self.spec = spec or WhisperSpec(80, 128, 2, 1, 1, 512, 2310)
        self.device = 0
        self.slots: List[FakeSlot] = []
        self.generate_script: List[Callable] = []
        self.default_tokens: List[int] = []
        self.lang_index = 0

    def create_slot(self, max_batch=1, rows=5):
        s = FakeSlot(self, max_batch, rows)
        self.slots.append(s)
        return s

    def script_generate(self, slot, prompts, ids, kw):
        if self.generate_script:
            return self.generate_script.pop(0)(prompts, ids, kw)
        return [GenerationResult([list(self.default_tokens)], [-0.1], 0.01) for _ in prompts]

    def script_lang(self, batch, lang_ids):
        p = np.full((batch, len(lang_ids)), 0.2 / max(1, len(lang_ids) - 1), np.float32)
        p[:, self.lang_index] = 0.8
        return p
