"""End-to-end transcripts from TRAINED weights (tests/golden/trained_tiny/, made by tests/golden/make_trained_tiny.py): a
Whisper-architecture checkpoint that Hugging Face `transformers` trained on a synthetic "tone language" until it transcribes
held-out utterances exactly. No OpenAI checkpoint exists offline and seeded random weights only produce noise tokens, so this is
the one place where a transcript can be RIGHT: the words that were played. For every held-out utterance

    ground truth  ==  Hugging Face's own beam search on the stored checkpoint (expected.json)
                  ==  this repo's host logic on the CPU oracle      (-m "not gpu")
                  ==  `WhisperModelHIP(path).transcribe(pcm)` on the MI355X through the C-ABI   (-m gpu)

i.e. word error rate 0 — the form the reference's only result-level test has (WER < 0.05 on assets/jfk.flac,
/root/reference/tests/test_server.py:73-118) — with the reference transcript produced by an independent implementation."""
import json
import os

import pytest

from tests import helpers as H
from tests.golden.make_trained_tiny import utterance

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_tiny")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(DIR, "model.safetensors")), reason="tests/golden/trained_tiny not generated")

# the reference's defaults (timestamps on, beam 5) with the quality fallbacks off (one decode per window)
KW = dict(language="en", beam_size=5, temperature=0.0, vad_filter=False, condition_on_previous_text=False,
          compression_ratio_threshold=None, log_prob_threshold=None, no_speech_threshold=None)


def _expected():
    with open(os.path.join(DIR, "expected.json")) as f:
        return json.load(f)


def _check(model, cases, what):
    exp = _expected()
    words, tb = exp["word_token_ids"], exp["timestamp_begin"]
    n_right = 0
    for c in cases:
        pcm, ws = utterance(c["seed"])
        assert ws == c["words"]
        assert c["hf_tokens"] == c["truth_tokens"], "fixture: Hugging Face itself must transcribe the held-out utterance exactly"
        segs, info = model.transcribe(pcm, **KW)
        assert len(segs) == 1, (what, [(s.start, s.end, s.text) for s in segs])
        toks = [t for t in segs[0].tokens if t < model.token_ids.eot]
        assert toks == [t for t in c["truth_tokens"] if t < tb], (what, c["seed"], [words.index(t) if t in words else t for t in toks], ws)
        assert segs[0].text.split() == [f"w{300 + w}" for w in ws], (what, segs[0].text)
        # the timestamps the model learned: the utterance starts at 0.00 and ends on the grid, 0.5 + 0.5 n seconds
        assert segs[0].start == 0.0 and abs(segs[0].end - (0.5 + 0.5 * len(ws))) < 1e-6, (what, segs[0].start, segs[0].end)
        # the score: sum of log-probs incl. the EOT (HF: sequence score x its length normaliser) = avg_logprob * (generated + 1)
        n_gen = len(c["truth_tokens"])
        got_sum = segs[0].avg_logprob * (n_gen + 1)
        assert abs(got_sum - c["hf_sum_logprob"]) <= 5e-2 + 2e-2 * abs(c["hf_sum_logprob"]), (what, got_sum, c["hf_sum_logprob"])
        assert info.language == "en" and abs(info.duration - 30.0) < 1e-6
        n_right += 1
    print(what, "word error rate 0 on", n_right, "held-out utterances (", sum(len(c["words"]) for c in cases), "words ), segment times exact")


def test_checkpoint_is_the_trained_one():
    from whisperlive_amd.specs import spec_from_state_dict
    from whisperlive_amd.weights import load_model_dir
    sd = load_model_dir(DIR)
    spec = spec_from_state_dict(sd)
    assert (spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers, spec.dec_layers, spec.ffn, spec.vocab) == (80, 128, 2, 2, 2, 512, 2310)
    exp = _expected()
    assert len(exp["cases"]) >= 8 and all(c["hf_tokens"] == c["truth_tokens"] and c["hf_ended_with_eot"] for c in exp["cases"])


def test_oracle_pipeline_transcribes_the_trained_checkpoint():
    from tests.oracle_engine import OracleEngine
    from whisperlive_amd.specs import spec_from_state_dict
    from whisperlive_amd.transcriber import WhisperModelHIP
    from whisperlive_amd.weights import load_model_dir
    sd = load_model_dir(DIR)
    model = WhisperModelHIP(DIR, engine=OracleEngine(spec_from_state_dict(sd), H.f16_weights(sd)))
    _check(model, _expected()["cases"][:4], "CPU oracle pipeline")


@pytest.mark.gpu
def test_hip_engine_transcribes_the_trained_checkpoint(gpu):
    from whisperlive_amd.transcriber import WhisperModelHIP
    model = WhisperModelHIP(DIR, device="cuda", device_index=0)          # loader -> spec -> tokenizer -> engine, all from the directory
    try:
        _check(model, _expected()["cases"], "MI355X (libwlx.so)")
        # the streaming session path on real (trained) speech: frames in, the words out
        import json as _json
        import time
        from unittest.mock import MagicMock
        from whisperlive_amd.serve_client import ServeClientHIP
        pcm, ws = utterance(_expected()["cases"][0]["seed"])
        n = int(16000 * (0.5 + 0.5 * len(ws) + 0.5))
        sock = MagicMock()
        c = ServeClientHIP(sock, client_uid="tt", model="x.en", transcriber=model, use_vad=False, same_output_threshold=2)
        for i in range(0, n, 4096):
            c.add_frames(pcm[i:i + 4096])
        deadline = time.time() + 20
        said = ""
        while time.time() < deadline:
            msgs = [_json.loads(a[0][0]) for a in sock.send.call_args_list]
            with_segs = [m for m in msgs if "segments" in m]
            said = " ".join(s["text"].strip() for s in with_segs[-1]["segments"]) if with_segs else ""   # the LATEST transcript message
            if said.split()[: len(ws)] == [f"w{300 + w}" for w in ws]:
                break
            time.sleep(0.1)
        c.cleanup(); c.trans_thread.join(timeout=5)
        assert said.split()[: len(ws)] == [f"w{300 + w}" for w in ws], said
    finally:
        model.close()
        model.engine.close()


# ---------------------------------------------------------------------------------------------------------------------------
# The same learned function through the BENCHMARKED kernels (VERDICT r03 'weak' 2 / task 3): the trained checkpoint widened
# function-preservingly (tests/golden/widen_trained_tiny.py) to Whisper-small's widths (d_model 768 / 12 heads / ffn 3072) and
# to large-v3's (1280 / 20 / 5120). d_model 128 is rejected by the lean decode kernels; 768 and 1280 are exactly their shapes.
LEAN_NAMES = ("dec_gemv2_kernel", "dec_self_attn2_kernel")


@pytest.fixture(scope="module")
def widened(tmp_path_factory):
    from tests.golden.widen_trained_tiny import widen_dir
    made = {}

    def get(r):
        if r not in made:
            made[r] = widen_dir(str(tmp_path_factory.mktemp(f"trained_wide{r}")), r)
        return made[r]
    return get


def test_widened_checkpoint_is_the_same_function_under_hugging_face(widened):
    """Hugging Face's own beam search on the widened (d_model 768) checkpoint returns the tokens it returned for d_model 128"""
    from tests.golden.widen_trained_tiny import hf_beam_tokens
    from whisperlive_amd.specs import spec_from_state_dict
    from whisperlive_amd.weights import load_model_dir
    d = widened(6)
    spec = spec_from_state_dict(load_model_dir(d))
    assert (spec.d_model, spec.n_heads, spec.ffn, spec.enc_layers, spec.dec_layers, spec.vocab) == (768, 12, 3072, 2, 2, 2310)
    cases = _expected()["cases"][:3]
    got = hf_beam_tokens(d, [c["seed"] for c in cases], threads=8)
    for c in cases:
        t, lp = got[c["seed"]]
        assert t == c["hf_tokens"] == c["truth_tokens"], (c["seed"], t)
        assert abs(lp - c["hf_sum_logprob"]) <= 2e-3 + 2e-2 * abs(c["hf_sum_logprob"]), (lp, c["hf_sum_logprob"])


@pytest.mark.gpu
@pytest.mark.parametrize("r,d_model", [(6, 768), (10, 1280)])
def test_hip_engine_transcribes_the_widened_checkpoint_through_the_lean_kernels(gpu, widened, r, d_model):
    """WhisperModelHIP(path).transcribe on the MI355X: word error rate 0 on the held-out utterances THROUGH the kernels bench.py
    times; the per-kernel profile of a 5-row decode step must list the lean kernels (and no first-generation dec_gemv_kernel)."""
    from whisperlive_amd.transcriber import WhisperModelHIP
    model = WhisperModelHIP(widened(r), device="cuda", device_index=0)
    try:
        assert model.engine.spec.d_model == d_model
        cases = _expected()["cases"]
        _check(model, cases if r == 6 else cases[:6], f"MI355X (libwlx.so), widened to d_model {d_model}")
        slot = model._slot()
        names = [k["name"] for k in slot.debug_profile_step(5, 8, 2)]
        print("decode-step kernels:", sorted(set(names)))
        assert not any(n.startswith("dec_gemv_kernel<") for n in names), names          # nothing fell back to the general kernel
        assert any(n.startswith("dec_gemv2_kernel<") for n in names) and any(n.startswith("dec_self_attn2_kernel") for n in names), names
        assert any(n.startswith("dec_cq_cross_attn_kernel") for n in names) == (d_model == 768), names   # the fused query + cross attention: Whisper-small shapes
        assert any(n.startswith("dec_gemv2_kernel<") and n.endswith(", 5, 1, 1, 0>") for n in names), names   # the K-split MLP projection (out mode 5 = GEMV_OUT_SLAB)
        assert any(n.startswith("dec_gemv2_kernel<") and n.endswith(", 1>") for n in names), names            # ... and a consumer of its slabs (xsrc 1)
        assert any(n.startswith("dec_vocab_kernel<") for n in names), names
    finally:
        model.close()
        model.engine.close()


@pytest.mark.gpu
def test_twelve_utterances_in_one_worker_batch_are_one_60_row_decode(gpu, widened):
    """The batching worker at `max_batch_size` 12 on the widened (Whisper-small-width) learned model: the 12 held-out utterances as ONE
    batch — per-item log-mel recorded and launched together, one batched encode, ONE beam-5 decode of 60 rows (the transcriber split
    batches at 48 rows until round 4: 9 + 3 clips) through the row-tiled lean kernels — every transcript word-exact."""
    from whisperlive_amd.batching import BatchInferenceWorker, BatchRequest
    from whisperlive_amd.transcriber import WhisperModelHIP
    model = WhisperModelHIP(widened(6), device="cuda", device_index=0, max_batch=12)
    try:
        cases = _expected()["cases"][:12]
        assert len(cases) == 12
        slot = model._slot()
        calls, orig = [], slot.generate

        def counted(prompts, *a, **k):
            calls.append(len(prompts))
            return orig(prompts, *a, **k)
        slot.generate = counted
        worker = BatchInferenceWorker(model, max_batch_size=12, batch_window_ms=10)
        worker.TEMPERATURES = (0.0,)
        reqs = [BatchRequest(audio=utterance(c["seed"])[0], language="en", use_vad=False) for c in cases]
        worker._process_multi(reqs)
        assert calls == [12], calls
        for c, r in zip(cases, reqs):
            assert r.error is None and r.future.is_set(), r.error
            said = " ".join(s.text.strip() for s in r.result).split()
            assert said == [f"w{300 + w}" for w in c["words"]], (c["seed"], said, c["words"])
        print("MI355X (libwlx.so), widened to d_model 768: 12 utterances in one worker batch = one 60-row decode, word error rate 0")
    finally:
        model.close()
        model.engine.close()
