"""Golden vectors for oracle/alignment.py from Hugging Face transformers (models/whisper/generation_whisper.py
_median_filter / _dynamic_time_warping — the same published algorithm ctranslate2 Whisper.align implements).
Run in the build container: python tests/golden/make_align_golden.py -> tests/golden/align_golden.npz"""
import os

import numpy as np
import torch
from transformers.models.whisper.generation_whisper import _dynamic_time_warping, _median_filter

rng = np.random.default_rng(17)
out = {}
for i, (n, m) in enumerate([(7, 40), (23, 311), (1, 9), (12, 5)]):
    x = rng.standard_normal((n, m)).astype(np.float32)
    ti, fi = _dynamic_time_warping(x.astype(np.float64))
    out[f"dtw_in_{i}"] = x
    out[f"dtw_ti_{i}"] = np.asarray(ti)
    out[f"dtw_fi_{i}"] = np.asarray(fi)
for i, (shape, w) in enumerate([((3, 5, 50), 7), ((2, 4, 9), 3), ((1, 2, 3), 7), ((2, 6, 120), 5)]):
    x = rng.standard_normal(shape).astype(np.float32)
    y = _median_filter(torch.from_numpy(x)[None], w)[0].numpy() if x.shape[-1] > w // 2 else x
    out[f"med_in_{i}"] = x
    out[f"med_w_{i}"] = np.asarray(w)
    out[f"med_out_{i}"] = y
np.savez_compressed(os.path.join(os.path.dirname(__file__), "align_golden.npz"), **out)
print("wrote", len(out), "arrays")
