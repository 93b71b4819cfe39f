#!/usr/bin/env python
"""Generate the golden fixtures that PIN the CPU oracle (oracle/) for the Whisper hot path.

Why this script exists: the reference's own tests pin no tensor on this path and its numerical
engines (faster-whisper 1.2.0, CTranslate2, onnxruntime) cannot be imported offline. The independent
implementation of the SAME published algorithm that IS importable in the build container is Hugging Face
`transformers` (Whisper model, WhisperFeatureExtractor, mel_filter_bank, WhisperTimeStampLogitsProcessor,
SuppressTokens*LogitsProcessor). This script runs those on seeded inputs and stores their outputs; tests/
(`-m "not gpu"`) then checks the oracle against the stored vectors WITHOUT needing transformers at test time.
All random inputs/weights come from numpy's PCG64 `default_rng(seed)` (stable across numpy versions), so the
fixtures hold only OUTPUTS plus the seeds.

Run (build container only):  python tests/golden/make_golden.py
Writes: tests/golden/hf_golden.npz (compressed, < 300 KB) + tests/golden/hf_golden.json (seeds / configs / tokens).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

MODEL_CASES = {
    # name: (n_mels, d_model, heads, enc_layers, dec_layers, ffn, vocab, weight seed, feature seed)
    "m64": (80, 64, 1, 1, 1, 128, 1700, 2024, 11),
    "m128": (128, 128, 2, 2, 2, 256, 1711, 2025, 12),
}


def np_weights(case):
    """Weights in HF state-dict naming from numpy's PCG64 (shared with tests/test_oracle_golden.py)."""
    from whisperlive_amd.specs import WhisperSpec
    from whisperlive_amd.weights import random_weights
    n_mels, d, H, le, ld, ffn, vocab, wseed, _ = MODEL_CASES[case]
    spec = WhisperSpec(n_mels, d, H, le, ld, ffn, vocab)
    return spec, random_weights(spec, seed=wseed)


def case_features(case):
    n_mels, *_rest, fseed = MODEL_CASES[case]
    return (np.random.default_rng(fseed).standard_normal((1, n_mels, 3000)) * 0.5).astype(np.float32)


def case_tokens(case, n=9):
    vocab = MODEL_CASES[case][6]
    return np.random.default_rng(MODEL_CASES[case][8] + 100).integers(0, vocab - 1501 - 120, size=n)


def token_layout(vocab):
    tb = vocab - 1501
    return dict(sot=tb - 106, eot=tb - 107, no_timestamps=tb - 1, timestamp_begin=tb, no_speech=tb - 2, blank=7)


def ts_cases(vocab):
    """(history, seed, peak) triples that fire every branch of the timestamp rules."""
    L = token_layout(vocab)
    tb = L["timestamp_begin"]
    return [
        ([], 1, 3.0), ([tb + 3], 2, 3.0), ([tb + 3, 17], 3, 3.0), ([tb + 3, 17, 25, tb + 40], 4, 3.0),
        ([tb + 3, 17, tb + 40, tb + 40], 5, 3.0), ([tb + 0, tb + 0], 6, 3.0), ([tb + 10, 5, 6, 7], 7, 0.5),
        ([tb + 1500], 8, 3.0), ([tb + 7, 9, tb + 1499, tb + 1499, 3], 9, 1.0),
    ]


def ts_scores(vocab, seed, peak):
    s = (np.random.default_rng(seed).standard_normal(vocab) * peak).astype(np.float32)
    s[vocab - 1501:] += np.float32(1.0 if seed % 2 else -1.0)
    return s


def main():
    import torch
    from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration
    from transformers.audio_utils import mel_filter_bank
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor,
                                                         SuppressTokensLogitsProcessor,
                                                         WhisperTimeStampLogitsProcessor)

    from oracle import logmel as olm

    out, meta = {}, {"transformers": __import__("transformers").__version__, "numpy": np.__version__}
    # ---- A. Slaney mel filterbanks
    for n in (80, 128):
        out[f"mel_filters_{n}"] = mel_filter_bank(201, n, 0.0, 8000.0, 16000, "slaney", "slaney").T.astype(np.float32)
    # ---- B. log-mel of a seeded clip through WhisperFeatureExtractor's numpy path (no 30 s padding, no 160 pad)
    pcm = olm.speech_like_pcm(2.5, seed=77)
    for n in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=n)
        out[f"logmel_{n}"] = np.asarray(fe._np_extract_fbank_features(pcm[None], "cpu"))[0].astype(np.float32)
    meta["logmel"] = dict(seconds=2.5, seed=77, note="oracle.log_mel_spectrogram(pcm, n, padding=0)")
    # ---- C. network: encoder states + teacher-forced logits on numpy-seeded weights
    for case in MODEL_CASES:
        spec, w = np_weights(case)
        cfg = WhisperConfig(vocab_size=spec.vocab, num_mel_bins=spec.n_mels, d_model=spec.d_model,
                            encoder_layers=spec.enc_layers, decoder_layers=spec.dec_layers,
                            encoder_attention_heads=spec.n_heads, decoder_attention_heads=spec.n_heads,
                            encoder_ffn_dim=spec.ffn, decoder_ffn_dim=spec.ffn, max_source_positions=1500,
                            max_target_positions=448, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                            decoder_start_token_id=1, activation_function="gelu", attn_implementation="eager")
        model = WhisperForConditionalGeneration(cfg).eval()
        sd = {k: torch.from_numpy(v) for k, v in w.items()}
        sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and all("k_proj.bias" not in m for m in missing), (missing, unexpected)
        feats = case_features(case)
        toks = case_tokens(case)
        with torch.no_grad():
            enc = model.model.encoder(torch.from_numpy(feats)).last_hidden_state
            logits = model(input_features=torch.from_numpy(feats), decoder_input_ids=torch.from_numpy(toks)[None]).logits
        out[f"{case}_enc_rows"] = enc[0, ::25].numpy().astype(np.float32)
        out[f"{case}_logits"] = logits[0].numpy().astype(np.float32)
        # ---- E. greedy decode with HF's own logits processors (timestamps on), no KV cache, 24 tokens
        L = token_layout(spec.vocab)
        suppress = sorted({1, 2, 5, L["sot"], L["timestamp_begin"] - 3, L["timestamp_begin"] - 4})

        class G:  # GenerationConfig stand-in with the fields the processor reads
            no_timestamps_token_id = L["no_timestamps"]
            eos_token_id = L["eot"]
            bos_token_id = L["eot"]
            max_initial_timestamp_index = 50
            _detect_timestamp_from_logprob = True

        prompt = [L["sot"]]
        procs = [SuppressTokensLogitsProcessor(suppress), SuppressTokensAtBeginLogitsProcessor([L["blank"], L["eot"]], len(prompt)),
                 WhisperTimeStampLogitsProcessor(G, begin_index=len(prompt))]
        seq, cum = list(prompt), 0.0
        gen = []
        with torch.no_grad():
            for _ in range(24):
                ids_t = torch.tensor([seq])
                sc = model(input_features=torch.from_numpy(feats), decoder_input_ids=ids_t).logits[:, -1].float()
                for p in procs:
                    sc = p(ids_t, sc)
                lp = torch.log_softmax(sc, dim=-1)
                tok = int(torch.argmax(lp[0]))
                cum += float(lp[0, tok])
                if tok == L["eot"]:
                    break
                gen.append(tok)
                seq.append(tok)
        meta[f"{case}_greedy"] = dict(prompt=prompt, suppress=suppress, tokens=gen, sum_logprob=cum, layout=L)
    # ---- D. timestamp-rule masks from WhisperTimeStampLogitsProcessor
    vocab = 1711
    L = token_layout(vocab)

    class G2:
        no_timestamps_token_id = L["no_timestamps"]
        eos_token_id = L["eot"]
        bos_token_id = L["eot"]
        max_initial_timestamp_index = 50
        _detect_timestamp_from_logprob = True

    proc = WhisperTimeStampLogitsProcessor(G2, begin_index=1)
    masks = []
    for hist, seed, peak in ts_cases(vocab):
        sc = torch.from_numpy(ts_scores(vocab, seed, peak))[None]
        res = proc(torch.tensor([[L["sot"]] + hist]), sc)
        masks.append(np.isneginf(res[0].numpy()))
    out["ts_masks"] = np.packbits(np.stack(masks), axis=1)
    meta["ts"] = dict(vocab=vocab, layout=L, n_cases=len(masks))
    np.savez_compressed(os.path.join(HERE, "hf_golden.npz"), **out)
    with open(os.path.join(HERE, "hf_golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(HERE, "hf_golden.npz")))


if __name__ == "__main__":
    main()
