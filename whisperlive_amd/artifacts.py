"""Where the model files come from: Whisper checkpoints by SIZE NAME or hub id, and the Silero VAD file.

The reference resolves both without the operator naming a path:

* a size name (`"small.en"`) or hub id goes to faster-whisper's `download_model` — the `Systran/faster-whisper-<size>` repositories,
  cache first, download otherwise (whisper_live/transcriber/transcriber_faster_whisper.py:620-632; names accepted:
  whisper_live/backend/faster_whisper_backend.py:74-79; other hub ids: `snapshot_download` + conversion, :133-178);
* Silero VAD is `~/.cache/whisper-live/silero_vad.onnx`, fetched on first use (whisper_live/vad.py:112-128); faster-whisper
  itself ships the ONNX inside its wheel (`faster_whisper/assets/`).

This module does the same LOOK-UP for the HIP backend — nothing here touches the GPU — so that `WhisperModelHIP("small.en")` and
`use_vad` work wherever the artefacts already are (a warmed Hugging Face cache, an installed faster-whisper wheel, the reference's
own cache directory) and download when the deployment allows it. Order, first hit wins:

  model:  an existing directory  ->  $WLX_MODEL_ROOT/<name>  ->  the Hugging Face cache (local_files_only) for every candidate
          repository of the name  ->  a download of the first candidate (unless local_files_only / HF_HUB_OFFLINE=1 /
          WLX_NO_DOWNLOAD=1)
  VAD:    $WLX_SILERO_VAD_NPZ / $WLX_SILERO_VAD_ONNX  ->  ~/.cache/whisper-live/silero_vad.onnx  ->  the ONNX files inside an
          installed faster_whisper / silero_vad package (located without importing them)  ->  a download to the reference's cache
          path (same conditions)

A directory counts as a model when the loaders can read it (whisperlive_amd/weights.py::load_model_dir): CTranslate2 `model.bin` or
Hugging Face `model.safetensors` (single or sharded), plus `tokenizer.json`.
"""
from __future__ import annotations

import glob
import importlib.util
import logging
import os
from typing import Callable, List, Optional, Sequence, Tuple

# the names the reference accepts (faster_whisper_backend.py:74-79) -> candidate repositories, the reference's own choice first
# (faster-whisper's published table), then the original OpenAI / distil-whisper checkpoints (Hugging Face safetensors, which
# load_model_dir reads as well)
_SYSTRAN = ("tiny", "tiny.en", "base", "base.en", "small", "small.en", "medium", "medium.en", "large-v1", "large-v2", "large-v3")
MODEL_REPOS = {n: [f"Systran/faster-whisper-{n}", f"openai/whisper-{n}"] for n in _SYSTRAN}
MODEL_REPOS.update({
    "large": ["Systran/faster-whisper-large-v3", "openai/whisper-large-v3"],
    "distil-small.en": ["Systran/faster-distil-whisper-small.en", "distil-whisper/distil-small.en"],
    "distil-medium.en": ["Systran/faster-distil-whisper-medium.en", "distil-whisper/distil-medium.en"],
    "distil-large-v2": ["Systran/faster-distil-whisper-large-v2", "distil-whisper/distil-large-v2"],
    "distil-large-v3": ["Systran/faster-distil-whisper-large-v3", "distil-whisper/distil-large-v3"],
    "large-v3-turbo": ["mobiuslabsgmbh/faster-whisper-large-v3-turbo", "openai/whisper-large-v3-turbo"],
    "turbo": ["mobiuslabsgmbh/faster-whisper-large-v3-turbo", "openai/whisper-large-v3-turbo"],
})
MODEL_SIZES = tuple(MODEL_REPOS)
_MODEL_FILES = ["config.json", "preprocessor_config.json", "generation_config.json", "model.bin", "model.safetensors",
                "model.safetensors.index.json", "model-*.safetensors", "tokenizer.json", "vocabulary.*"]

SILERO_URL = "https://github.com/snakers4/silero-vad/raw/v5.0/files/silero_vad.onnx"      # whisper_live/vad.py:112
SILERO_CACHE = os.path.join("~", ".cache", "whisper-live", "silero_vad.onnx")              # whisper_live/vad.py:113-119


class ArtifactNotFound(FileNotFoundError):
    """nothing usable at any of the places looked at; the message lists them"""


def is_model_dir(path: str) -> bool:
    if not (path and os.path.isdir(path)):
        return False
    has_w = (os.path.isfile(os.path.join(path, "model.bin")) or os.path.isfile(os.path.join(path, "model.safetensors"))
             or os.path.isfile(os.path.join(path, "model.safetensors.index.json")))
    return has_w and os.path.isfile(os.path.join(path, "tokenizer.json"))


def downloads_allowed(local_files_only: Optional[bool] = None) -> bool:
    if local_files_only:
        return False
    off = lambda k: os.environ.get(k, "").strip().lower() in ("1", "true", "yes", "on")
    return not (off("HF_HUB_OFFLINE") or off("WLX_NO_DOWNLOAD"))


def candidate_repos(name: str) -> List[str]:
    if name in MODEL_REPOS:
        return list(MODEL_REPOS[name])
    if "/" in name and not name.startswith((".", "/", "~")):
        return [name]                                   # a hub id (faster_whisper_backend.py:139-150)
    return []


def _hub_snapshot(repo: str, cache_dir: Optional[str], local_only: bool) -> Optional[str]:
    try:
        from huggingface_hub import snapshot_download
    except ImportError:
        return None
    try:
        return snapshot_download(repo_id=repo, repo_type="model", cache_dir=cache_dir, local_files_only=local_only,
                                 allow_patterns=_MODEL_FILES, etag_timeout=5)
    except Exception as e:  # noqa: BLE001 — not cached / no network / gated: the caller moves on to the next place
        logging.debug("hub %s (%s): %s: %s", repo, "cache" if local_only else "download", type(e).__name__, e)
        return None


def resolve_model(name_or_path: str, download_root: Optional[str] = None, local_files_only: Optional[bool] = None,
                  snapshot: Optional[Callable[[str, Optional[str], bool], Optional[str]]] = None) -> str:
    """A model DIRECTORY for `name_or_path` (see the module docstring for the order). `snapshot` replaces the hub call in tests."""
    snap = snapshot or _hub_snapshot
    tried: List[str] = []
    p = os.path.expanduser(str(name_or_path))
    if os.path.isdir(p):
        return p
    tried.append(f"directory {p}")
    root = os.environ.get("WLX_MODEL_ROOT")
    if root:
        for cand in (os.path.join(root, name_or_path), os.path.join(root, f"faster-whisper-{name_or_path}"), os.path.join(root, f"whisper-{name_or_path}")):
            if is_model_dir(cand):
                return cand
        tried.append(f"$WLX_MODEL_ROOT={root}")
    repos = candidate_repos(str(name_or_path))
    if not repos:
        raise ArtifactNotFound(f"'{name_or_path}' is neither a model directory, a known size ({', '.join(MODEL_SIZES)}) nor a hub id; looked at: "
                               + "; ".join(tried))
    for repo in repos:                                   # cache first, every candidate
        d = snap(repo, download_root, True)
        if d and is_model_dir(d):
            logging.info("model '%s': cached snapshot of %s at %s", name_or_path, repo, d)
            return d
        tried.append(f"Hugging Face cache: {repo}")
    if downloads_allowed(local_files_only):
        for repo in repos:
            d = snap(repo, download_root, False)
            if d and is_model_dir(d):
                logging.info("model '%s': downloaded %s to %s", name_or_path, repo, d)
                return d
            tried.append(f"download: {repo}")
    else:
        tried.append("download: not allowed (local_files_only / HF_HUB_OFFLINE / WLX_NO_DOWNLOAD)")
    raise ArtifactNotFound(f"no Whisper checkpoint found for '{name_or_path}' — pass a CTranslate2 (model.bin) or Hugging Face "
                           f"(model.safetensors) directory, set WLX_MODEL_ROOT, or warm the Hugging Face cache. Looked at: " + "; ".join(tried))


def _package_files(package: str, patterns: Sequence[str]) -> List[str]:
    """files inside an installed package, found WITHOUT importing it (faster_whisper pulls in ctranslate2 / onnxruntime)"""
    try:
        spec = importlib.util.find_spec(package)
    except (ImportError, ValueError):
        return []
    out: List[str] = []
    for loc in (list(spec.submodule_search_locations) if spec and spec.submodule_search_locations else []):
        for pat in patterns:
            out.extend(sorted(glob.glob(os.path.join(loc, pat))))
    return out


def silero_candidates() -> List[Tuple[str, str]]:
    """(kind, path) in look-up order; kind is "npz" or "onnx". Files that do not exist are left out."""
    out: List[Tuple[str, str]] = []
    for env, kind in (("WLX_SILERO_VAD_NPZ", "npz"), ("WLX_SILERO_VAD_ONNX", "onnx")):
        v = os.environ.get(env)
        if v and os.path.isfile(os.path.expanduser(v)):
            out.append((kind, os.path.expanduser(v)))
    c = os.path.expanduser(SILERO_CACHE)
    if os.path.isfile(c):
        out.append(("onnx", c))
    for f in _package_files("faster_whisper", ["assets/silero_vad*.onnx", "assets/silero*.onnx"]) + \
            _package_files("silero_vad", ["data/silero_vad.onnx", "data/silero_vad*.onnx"]):
        if ("onnx", f) not in out:
            out.append(("onnx", f))
    return out


def download_silero(fetch: Optional[Callable[[str, str], None]] = None) -> Optional[str]:
    """the reference's first-use download (vad.py:112-128), to the same cache path; None when it fails"""
    target = os.path.expanduser(SILERO_CACHE)
    try:
        os.makedirs(os.path.dirname(target), exist_ok=True)
        if fetch is not None:
            fetch(SILERO_URL, target)
        else:
            import urllib.request
            with urllib.request.urlopen(SILERO_URL, timeout=10) as r, open(target + ".part", "wb") as f:
                f.write(r.read())
            os.replace(target + ".part", target)
        return target if os.path.isfile(target) else None
    except Exception as e:  # noqa: BLE001
        logging.debug("silero download: %s: %s", type(e).__name__, e)
        return None
