#!/usr/bin/env python
"""Golden vectors that pin the oracle's BEAM SEARCH (oracle/decoding.py, the decision procedure search.hip implements) to
code nobody here wrote: Hugging Face `GenerationMixin.generate(num_beams=N, early_stopping=True, length_penalty=1,
num_return_sequences=N)` on the two seeded golden models of make_golden.py, with HF's own Whisper logits processors
(SuppressTokens, SuppressTokensAtBegin, WhisperTimeStampLogitsProcessor).

Two adaptors make the comparison about the SEARCH and nothing else, both applied on the HF side as logits processors:
  * `Bias` — a seeded per-token logit offset (incl. +0 / +4 / +7 on <|endoftext|>): seeded random weights never emit EOT
    on their own, and without finished hypotheses the early-stopping / hypothesis-registration half of the search would go
    untested;
  * `Renorm` — log-softmax AFTER the masks. HF's beam search adds log-probs normalised BEFORE its processors mask ids;
    CTranslate2 (and OpenAI's reference decoder) mask logits first and normalise after, which is what the oracle does.
    Without this the cumulative scores differ by the log of the masked mass and beams are ranked differently.
What is stored: for every case the N returned hypotheses (tokens without the EOT) and their SUM of log-probs (HF's
sequence score x HF's length normaliser). The one remaining, deliberate difference is that normaliser: HF divides by the
generated length INCLUDING the EOT; the reference's own code (whisper_live/transcriber/transcriber_faster_whisper.py
:1412-1414 recovers the sum as score * len(tokens) ** length_penalty, tokens WITHOUT the EOT) fixes CTranslate2's as
EXCLUDING it, and that is what the oracle implements.

Run (build container only):  python tests/golden/make_beam_golden.py   -> tests/golden/beam_golden.json (~20 KB)
"""
from __future__ import annotations

import json
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import make_golden as mg  # noqa: E402

N_SEEDS = 10


def case_inputs(case: str, seed: int):
    """(features, logit bias, max_new_tokens, num_beams) of beam case (model, seed) — shared with the test"""
    n_mels, vocab = mg.MODEL_CASES[case][0], mg.MODEL_CASES[case][6]
    L = mg.token_layout(vocab)
    rng = np.random.default_rng(500 + seed)
    feats = (rng.standard_normal((1, n_mels, 3000)) * 0.5).astype(np.float32)
    bias = (rng.standard_normal(vocab) * 1.5).astype(np.float32)
    bias[L["eot"]] += [0.0, 4.0, 7.0][seed % 3]
    max_new = int(rng.integers(8, 25))
    beams = 3 if seed % 5 == 4 else 5
    return feats, bias, max_new, beams


def suppress_ids(L):
    return sorted({1, 2, 5, L["sot"], L["timestamp_begin"] - 3, L["timestamp_begin"] - 4})


def main():
    import torch
    from transformers import GenerationConfig, WhisperConfig, WhisperForConditionalGeneration
    from transformers.generation import GenerationMixin, LogitsProcessor, LogitsProcessorList
    from transformers.generation.logits_process import (SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor,
                                                         WhisperTimeStampLogitsProcessor)
    logging.disable(logging.WARNING)

    class Bias(LogitsProcessor):
        def __init__(self, b):
            self.b = torch.from_numpy(b)

        def __call__(self, ids, scores):
            return scores + self.b

    class Renorm(LogitsProcessor):
        def __call__(self, ids, scores):
            return torch.log_softmax(scores.float(), dim=-1)

    out = {"transformers": __import__("transformers").__version__, "cases": []}
    for case in mg.MODEL_CASES:
        spec, w = mg.np_weights(case)
        cfg = WhisperConfig(vocab_size=spec.vocab, num_mel_bins=spec.n_mels, d_model=spec.d_model,
                            encoder_layers=spec.enc_layers, decoder_layers=spec.dec_layers,
                            encoder_attention_heads=spec.n_heads, decoder_attention_heads=spec.n_heads,
                            encoder_ffn_dim=spec.ffn, decoder_ffn_dim=spec.ffn, max_source_positions=1500,
                            max_target_positions=448, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                            decoder_start_token_id=1, activation_function="gelu", attn_implementation="eager")
        model = WhisperForConditionalGeneration(cfg).eval()
        sd = {k: torch.from_numpy(v) for k, v in w.items()}
        sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
        model.load_state_dict(sd, strict=False)
        L = mg.token_layout(spec.vocab)

        class G:  # GenerationConfig stand-in with the fields the timestamp processor reads
            no_timestamps_token_id = L["no_timestamps"]
            eos_token_id = L["eot"]
            bos_token_id = L["eot"]
            max_initial_timestamp_index = 50
            _detect_timestamp_from_logprob = True

        for seed in range(N_SEEDS):
            feats, bias, max_new, beams = case_inputs(case, seed)
            procs = LogitsProcessorList([Bias(bias), SuppressTokensLogitsProcessor(suppress_ids(L)),
                                         SuppressTokensAtBeginLogitsProcessor([L["blank"], L["eot"]], 1),
                                         WhisperTimeStampLogitsProcessor(G, begin_index=1), Renorm()])
            gc = GenerationConfig(num_beams=beams, num_return_sequences=beams, early_stopping=True, length_penalty=1.0,
                                  max_new_tokens=max_new, do_sample=False, eos_token_id=L["eot"], pad_token_id=L["eot"],
                                  decoder_start_token_id=L["sot"], bos_token_id=L["sot"], output_scores=True,
                                  return_dict_in_generate=True)
            with torch.no_grad():
                r = GenerationMixin.generate(model, input_features=torch.from_numpy(feats),
                                             decoder_input_ids=torch.tensor([[L["sot"]]]), generation_config=gc,
                                             logits_processor=procs)
            hyps = []
            for s, sc in zip(r.sequences.tolist(), r.sequences_scores.tolist()):
                t = s[1:]
                ended = L["eot"] in t
                if ended:
                    t = t[: t.index(L["eot"])]
                hf_len = len(t) + (1 if ended else 0)            # HF normalises by the generated length incl. the EOT
                hyps.append(dict(tokens=t, ended_with_eot=ended, hf_score=sc, sum_logprob=sc * hf_len))
            out["cases"].append(dict(model=case, seed=seed, max_new_tokens=max_new, num_beams=beams, hypotheses=hyps))
            print(case, seed, beams, max_new, [len(h["tokens"]) for h in hyps], [h["ended_with_eot"] for h in hyps])
    with open(os.path.join(HERE, "beam_golden.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", os.path.getsize(os.path.join(HERE, "beam_golden.json")), "bytes")


if __name__ == "__main__":
    main()
