"""The Hugging Face checkpoint path of the loader (whisperlive_amd/weights.py load_hf_dir / load_model_dir) against a
directory written by transformers ITSELF (`WhisperForConditionalGeneration.save_pretrained`): file format (safetensors,
single file and sharded with an index), key naming (`model.` prefix, the tied `proj_out.weight` that must not be loaded
twice), architecture inference from shapes, and — through the CPU oracle — the numbers: the logits of the loaded state
dict equal the logits of the HF model that wrote the files. This is an external pin of the loader; the CTranslate2
model.bin path has none offline (tests/test_ct2_loader.py round-trips its own writer)."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

from oracle import model as omodel                                   # noqa: E402
from whisperlive_amd.specs import spec_from_state_dict              # noqa: E402
from whisperlive_amd.weights import load_model_dir                  # noqa: E402


def _hf_model(seed=0):
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    torch.manual_seed(seed)
    cfg = WhisperConfig(vocab_size=2310, num_mel_bins=80, d_model=128, encoder_layers=2, decoder_layers=3,
                        encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=512, decoder_ffn_dim=512,
                        max_source_positions=1500, max_target_positions=448, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                        decoder_start_token_id=1, activation_function="gelu", attn_implementation="eager")
    return WhisperForConditionalGeneration(cfg).eval()


@pytest.mark.parametrize("sharded", [False, True])
def test_directory_written_by_transformers_loads_and_computes_the_same_logits(tmp_path, sharded):
    model = _hf_model()
    d = str(tmp_path / "ckpt")
    model.save_pretrained(d, safe_serialization=True, **({"max_shard_size": "1MB"} if sharded else {}))
    files = sorted(os.listdir(d))
    assert ("model.safetensors.index.json" in files) == sharded, files
    sd = load_model_dir(d)
    assert "proj_out.weight" not in sd and all(k.startswith("model.") for k in sd)
    spec = spec_from_state_dict(sd)
    assert (spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers, spec.dec_layers, spec.ffn, spec.vocab) == (80, 128, 2, 2, 3, 512, 2310)
    ref = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        assert k in ref and np.array_equal(v, ref[k]), k
    assert set(ref) - set(sd) <= {"proj_out.weight"}
    # numbers: the oracle on the LOADED weights == the HF model that wrote them
    rng = np.random.default_rng(3)
    feats = (rng.standard_normal((1, 80, 3000)) * 0.5).astype(np.float32)
    toks = rng.integers(0, 700, size=(1, 7))
    with torch.no_grad():
        want = model(input_features=torch.from_numpy(feats), decoder_input_ids=torch.from_numpy(toks)).logits[0].numpy()
    oracle = omodel.WhisperOracle(omodel.Spec(spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers, spec.dec_layers,
                                              spec.ffn, spec.vocab), sd)
    got = oracle.decode_logits(oracle.encode(feats), toks)[0].numpy()
    assert np.abs(got - want).max() <= 5e-4 * max(1.0, np.abs(want).max())


def test_config_json_of_the_written_directory_matches_the_inferred_spec(tmp_path):
    model = _hf_model(1)
    d = str(tmp_path / "ckpt")
    model.save_pretrained(d, safe_serialization=True)
    with open(os.path.join(d, "config.json")) as f:
        cfg = json.load(f)
    spec = spec_from_state_dict(load_model_dir(d))
    assert (cfg["d_model"], cfg["encoder_layers"], cfg["decoder_layers"], cfg["vocab_size"], cfg["num_mel_bins"]) == (
        spec.d_model, spec.enc_layers, spec.dec_layers, spec.vocab, spec.n_mels)
