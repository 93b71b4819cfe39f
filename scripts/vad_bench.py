#!/usr/bin/env python
"""Silero-VAD probabilities for one 30 s chunk (938 windows): MI355X kernels (wlx_vad_probs) beside the CPU
restatement (oracle/silero_vad.py, fp32 numpy, the 'port' baseline — the reference runs this network through
onnxruntime on one CPU thread). Prints one JSON line."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from oracle import logmel as olm          # noqa: E402  (checker / baseline leg only)
from oracle import silero_vad as sv       # noqa: E402
from whisperlive_amd import vad           # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    w = sv.random_weights(3)
    pcm = olm.speech_like_pcm(seconds, seed=7)
    m = vad.SileroHIPModel(w, device=0)
    for _ in range(5):
        got = m(pcm)
    dev, wall = [], []
    for _ in range(50):
        t0 = time.perf_counter()
        m(pcm)
        wall.append((time.perf_counter() - t0) * 1e3)
        dev.append(m.last_device_ms)
    import torch
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    want = sv.speech_probs(w, pcm, np.float32)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    flop = got.shape[0] * 2 * (258 * 256 * 4 + 128 * 129 * 3 * 4 + 64 * 128 * 3 * 2 + 64 * 64 * 2 + 128 * 64 + 512 * 128 * 2)
    print(json.dumps({"workload": f"silero-vad probs, {seconds:g} s chunk, {got.shape[0]} windows", "dtype": "f32",
                      "device_ms_p50": float(np.median(dev)), "call_ms_p50": float(np.median(wall)),
                      "cpu_port_ms": cpu_ms, "cpu_cores": 1, "max_abs_err": float(np.abs(got - want).max()),
                      "algorithmic_gflop": flop / 1e9, "audio_s_per_s": seconds / (np.median(wall) / 1e3)}))


if __name__ == "__main__":
    main()
