"""Whisper architecture table: n_audio_ctx=1500, n_text_ctx=448, head_dim=64 everywhere."""
from __future__ import annotations

from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class WhisperSpec:
    n_mels: int
    d_model: int
    n_heads: int
    enc_layers: int
    dec_layers: int
    ffn: int
    vocab: int
    n_audio_ctx: int = 1500
    n_text_ctx: int = 448

    @property
    def multilingual(self) -> bool:
        return self.vocab >= 51865

    def asdict(self):
        return asdict(self)


def _s(n_mels, d, layers, vocab, dec_layers=None):
    return WhisperSpec(n_mels, d, d // 64, layers, dec_layers if dec_layers is not None else layers, 4 * d, vocab)


SPECS = {
    "tiny.en": _s(80, 384, 4, 51864), "tiny": _s(80, 384, 4, 51865),
    "base.en": _s(80, 512, 6, 51864), "base": _s(80, 512, 6, 51865),
    "small.en": _s(80, 768, 12, 51864), "small": _s(80, 768, 12, 51865),
    "medium.en": _s(80, 1024, 24, 51864), "medium": _s(80, 1024, 24, 51865),
    "large-v2": _s(80, 1280, 32, 51865), "large-v3": _s(128, 1280, 32, 51866),
    "large-v3-turbo": _s(128, 1280, 32, 51866, dec_layers=4), "turbo": _s(128, 1280, 32, 51866, dec_layers=4),
}


def get_spec(name: str) -> WhisperSpec:
    key = name.split("/")[-1].replace("whisper-", "").replace("faster-", "")
    if key not in SPECS:
        raise KeyError(f"unknown Whisper model size '{name}' (known: {sorted(SPECS)})")
    return SPECS[key]


def spec_from_state_dict(sd) -> WhisperSpec:
    """Infer the architecture from Hugging Face state-dict shapes."""
    d, n_mels, _ = sd["model.encoder.conv1.weight"].shape
    vocab = sd["model.decoder.embed_tokens.weight"].shape[0]
    ffn = sd["model.encoder.layers.0.fc1.weight"].shape[0]
    enc = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("model.encoder.layers."))
    dec = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("model.decoder.layers."))
    return WhisperSpec(int(n_mels), int(d), int(d) // 64, enc, dec, int(ffn), int(vocab))
