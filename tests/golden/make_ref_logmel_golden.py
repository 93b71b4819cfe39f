"""Generate tests/golden/ref_logmel_golden.npz: log-mel vectors produced BY THE REFERENCE'S OWN CODE.

Run in the build container (needs /root/reference):  python tests/golden/make_ref_logmel_golden.py

The reference's in-tree log-mel (whisper_live/transcriber/tensorrt_utils.py:130-194, plain torch-CPU) is loaded by path with
its un-vendored imports stubbed (as tests/test_reference_logmel_diff.py does) and evaluated with ``padding=160`` — the call the
faster-whisper path makes (transcriber_faster_whisper.py:862) — on seeded PCM (whisperlive_amd.synthetic.speech_like_pcm). Its
mel_filters.npz is not in the checkout; the filterbank written for it is oracle.logmel.mel_filters (== transformers'
mel_filter_bank to 1 ulp, tests/test_oracle_golden.py). Only DATA is stored: (seconds, seed, n_mels) -> float16-free fp32 map.
The fixture lets the GPU box — which has no /root/reference — check the HIP log-mel against vectors the reference computed."""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import logmel as olm  # noqa: E402
from whisperlive_amd.synthetic import speech_like_pcm  # noqa: E402

CASES = [(0.27, 11, 80), (1.0, 12, 80), (3.7, 13, 80), (1.0, 14, 128), (2.5, 15, 128)]     # (seconds, seed, n_mels)


def load_reference():
    for name in ("kaldialign", "soundfile", "av", "whisper_live", "whisper_live.utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["whisper_live"].__path__ = []
    sys.modules["whisper_live.utils"].resample = None
    spec = importlib.util.spec_from_file_location("_ref_tensorrt_utils", "/root/reference/whisper_live/transcriber/tensorrt_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    d = tempfile.mkdtemp()
    np.savez_compressed(os.path.join(d, "mel_filters.npz"), mel_80=olm.mel_filters(80), mel_128=olm.mel_filters(128))
    out = {"cases": np.asarray(CASES, np.float64)}
    for i, (sec, seed, n_mels) in enumerate(CASES):
        pcm = speech_like_pcm(sec, seed=seed)
        with torch.no_grad():
            out[f"logmel_{i}"] = ref.log_mel_spectrogram(torch.from_numpy(pcm), int(n_mels), padding=160, mel_filters_dir=d).numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_logmel_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
