"""Model-name and VAD-asset resolution (whisperlive_amd/artifacts.py) — the look-up the reference does through faster-whisper's
`download_model` / `snapshot_download` (whisper_live/transcriber/transcriber_faster_whisper.py:620-632,
whisper_live/backend/faster_whisper_backend.py:74-79,133-178) and `VoiceActivityDetection.download` (whisper_live/vad.py:112-128):
the ORDER of places, with fake cache directories; nothing here needs a GPU or a network."""
import os

import numpy as np
import pytest

from whisperlive_amd import artifacts as A


def _fake_model_dir(root, name, kind="ct2"):
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "model.bin" if kind == "ct2" else "model.safetensors"), "wb").close()
    open(os.path.join(d, "tokenizer.json"), "w").close()
    return d


def test_the_references_size_names_are_known():
    # faster_whisper_backend.py:74-79
    for n in ["tiny", "tiny.en", "base", "base.en", "small", "small.en", "medium", "medium.en", "large-v2", "large-v3", "distil-small.en",
              "distil-medium.en", "distil-large-v2", "distil-large-v3", "large-v3-turbo", "turbo"]:
        repos = A.candidate_repos(n)
        assert repos and all("/" in r for r in repos), n
    assert A.candidate_repos("small.en")[0] == "Systran/faster-whisper-small.en"        # the reference's own choice first
    assert A.candidate_repos("someone/whisper-finetune") == ["someone/whisper-finetune"]     # a hub id is taken as it is
    assert A.candidate_repos("no-such-size") == [] and A.candidate_repos("./relative/dir") == []


def test_resolution_order(tmp_path, monkeypatch):
    calls = []
    cache = {}

    def snap(repo, cache_dir, local_only):
        calls.append((repo, local_only))
        return cache.get((repo, local_only))

    monkeypatch.delenv("WLX_MODEL_ROOT", raising=False)
    monkeypatch.delenv("WLX_NO_DOWNLOAD", raising=False)
    monkeypatch.delenv("HF_HUB_OFFLINE", raising=False)
    # 1. an existing directory wins, the hub is never asked
    d = _fake_model_dir(str(tmp_path), "mine")
    assert A.resolve_model(d, snapshot=snap) == d and calls == []
    # 2. $WLX_MODEL_ROOT/<name>, also under the repository's own directory name
    root = str(tmp_path / "root"); os.makedirs(root)
    monkeypatch.setenv("WLX_MODEL_ROOT", root)
    r1 = _fake_model_dir(root, "faster-whisper-small.en")
    assert A.resolve_model("small.en", snapshot=snap) == r1 and calls == []
    r2 = _fake_model_dir(root, "small.en", kind="hf")
    assert A.resolve_model("small.en", snapshot=snap) == r2
    # 3. the Hugging Face cache, every candidate repository, BEFORE any download
    monkeypatch.delenv("WLX_MODEL_ROOT")
    hf = _fake_model_dir(str(tmp_path), "snap-openai", kind="hf")
    cache[("openai/whisper-base.en", True)] = hf
    assert A.resolve_model("base.en", snapshot=snap) == hf
    assert calls == [("Systran/faster-whisper-base.en", True), ("openai/whisper-base.en", True)]
    # 4. nothing cached: a download of the candidates in order ...
    calls.clear()
    dl = _fake_model_dir(str(tmp_path), "snap-dl")
    cache[("Systran/faster-whisper-tiny.en", False)] = dl
    assert A.resolve_model("tiny.en", snapshot=snap) == dl
    assert calls == [("Systran/faster-whisper-tiny.en", True), ("openai/whisper-tiny.en", True), ("Systran/faster-whisper-tiny.en", False)]
    # ... unless the caller / the environment forbids it; the error names every place looked at
    for how in ("arg", "HF_HUB_OFFLINE", "WLX_NO_DOWNLOAD"):
        calls.clear()
        if how != "arg":
            monkeypatch.setenv(how, "1")
        with pytest.raises(A.ArtifactNotFound) as ei:
            A.resolve_model("tiny.en", snapshot=snap, local_files_only=(how == "arg") or None)
        assert all(lo for _, lo in calls) and "Systran/faster-whisper-tiny.en" in str(ei.value) and "not allowed" in str(ei.value)
        monkeypatch.delenv(how, raising=False)
    # a snapshot that is not a loadable model directory (no tokenizer.json) is not accepted
    bad = str(tmp_path / "bad"); os.makedirs(bad); open(os.path.join(bad, "model.bin"), "wb").close()
    cache[("Systran/faster-whisper-medium", True)] = bad
    with pytest.raises(A.ArtifactNotFound):
        A.resolve_model("medium", snapshot=snap, local_files_only=True)
    with pytest.raises(A.ArtifactNotFound, match="neither a model directory"):
        A.resolve_model("no-such-size", snapshot=snap)
    assert issubclass(A.ArtifactNotFound, FileNotFoundError)


def test_transcriber_resolves_a_size_name_through_the_lookup(tmp_path, monkeypatch):
    """`WhisperModelHIP("small.en")` used to raise unless it was a directory: now it goes through resolve_model, and the error of an empty
    look-up is a FileNotFoundError that lists the places (the server turns it into the client's ERROR message as before)."""
    from whisperlive_amd.transcriber import WhisperModelHIP
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    monkeypatch.setenv("WLX_MODEL_ROOT", str(tmp_path / "empty"))
    with pytest.raises(FileNotFoundError, match="Systran/faster-whisper-small.en"):
        WhisperModelHIP("small.en")


def test_silero_lookup_order(tmp_path, monkeypatch):
    home = tmp_path / "home"
    monkeypatch.setenv("HOME", str(home))
    monkeypatch.delenv("WLX_SILERO_VAD_NPZ", raising=False)
    monkeypatch.delenv("WLX_SILERO_VAD_ONNX", raising=False)
    monkeypatch.setattr(A, "_package_files", lambda pkg, pats: [])
    assert A.silero_candidates() == []
    # the reference's own cache file (whisper_live/vad.py:113-119)
    c = home / ".cache" / "whisper-live"; c.mkdir(parents=True)
    (c / "silero_vad.onnx").write_bytes(b"x")
    assert A.silero_candidates() == [("onnx", str(c / "silero_vad.onnx"))]
    # an installed faster-whisper wheel's bundled file comes after it, the two variables before it
    pk = tmp_path / "site" / "faster_whisper" / "assets"; pk.mkdir(parents=True)
    (pk / "silero_vad_v6.onnx").write_bytes(b"y")
    monkeypatch.setattr(A, "_package_files", lambda pkg, pats: [str(pk / "silero_vad_v6.onnx")] if pkg == "faster_whisper" else [])
    npz = tmp_path / "w.npz"; np.savez(npz, a=np.zeros(1))
    monkeypatch.setenv("WLX_SILERO_VAD_NPZ", str(npz))
    monkeypatch.setenv("WLX_SILERO_VAD_ONNX", str(tmp_path / "missing.onnx"))          # a variable naming no file is skipped
    assert A.silero_candidates() == [("npz", str(npz)), ("onnx", str(c / "silero_vad.onnx")), ("onnx", str(pk / "silero_vad_v6.onnx"))]
    # first-use download goes to the reference's cache path
    (c / "silero_vad.onnx").unlink()
    got = A.download_silero(fetch=lambda url, target: open(target, "wb").write(b"z") and None)
    assert got == str(c / "silero_vad.onnx") and "silero-vad" in A.SILERO_URL
    assert A.download_silero(fetch=lambda url, target: (_ for _ in ()).throw(OSError("no network"))) in (None, got)


def test_package_files_does_not_import_the_package(tmp_path, monkeypatch):
    import sys
    site = tmp_path / "site"; pk = site / "wlx_fake_pkg" / "assets"; pk.mkdir(parents=True)
    (site / "wlx_fake_pkg" / "__init__.py").write_text("raise RuntimeError('must not be imported')\n")
    (pk / "silero_vad.onnx").write_bytes(b"x")
    monkeypatch.syspath_prepend(str(site))
    assert A._package_files("wlx_fake_pkg", ["assets/silero*.onnx"]) == [str(pk / "silero_vad.onnx")]
    assert "wlx_fake_pkg" not in sys.modules
    assert A._package_files("wlx_no_such_pkg", ["*"]) == []


def test_default_vad_model_skips_a_file_that_is_not_silero(tmp_path, monkeypatch):
    """a candidate that does not parse as the Silero stack is reported and skipped; with nothing usable use_vad stays unavailable"""
    from whisperlive_amd import vad
    home = tmp_path / "home"; c = home / ".cache" / "whisper-live"; c.mkdir(parents=True)
    (c / "silero_vad.onnx").write_bytes(b"not an onnx file")
    monkeypatch.setenv("HOME", str(home))
    for k in ("WLX_SILERO_VAD_NPZ", "WLX_SILERO_VAD_ONNX", "WLX_ALLOW_VAD_STANDIN"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(A, "_package_files", lambda pkg, pats: [])
    monkeypatch.setattr(vad, "_default_model", None)
    monkeypatch.setattr(vad, "_default_weights", None)
    monkeypatch.setattr(vad, "_device_models", {})
    monkeypatch.setattr(vad, "_resolve_failed", None)
    with pytest.raises(vad.VadUnavailable, match="whisper-live/silero_vad.onnx"):
        vad.get_default_model(0)


def test_default_vad_lookup_runs_once_and_outside_the_model_lock(monkeypatch):
    """the look-up (which may download for up to 10 s) ran under the process-wide model lock on EVERY call that found no
    weights. Now: one look-up per process, without the model lock; a failed one is remembered and later calls raise at once;
    set_default_model(None) / configure() forget the failure."""
    from whisperlive_amd import vad
    for k in ("WLX_SILERO_VAD_NPZ", "WLX_SILERO_VAD_ONNX", "WLX_ALLOW_VAD_STANDIN"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(vad, "_default_model", None)
    monkeypatch.setattr(vad, "_default_weights", None)
    monkeypatch.setattr(vad, "_device_models", {})
    monkeypatch.setattr(vad, "_resolve_failed", None)
    calls = []

    def finder():
        calls.append(vad._default_lock.acquire(blocking=False))      # the model lock is FREE while the look-up runs
        if calls[-1]:
            vad._default_lock.release()
        return None
    monkeypatch.setattr(vad, "_find_default_weights", finder)
    for _ in range(3):
        with pytest.raises(vad.VadUnavailable):
            vad.get_default_model(0)
    assert calls == [True]                                            # looked up once, lock not held; calls 2 and 3 raised immediately
    vad.set_default_model(None)
    with pytest.raises(vad.VadUnavailable):
        vad.get_default_model(0)
    assert calls == [True, True]
    monkeypatch.setenv("WLX_ALLOW_VAD_STANDIN", "1")
    assert isinstance(vad.get_default_model(0), vad.EnergyGateModel)
    vad.set_default_model(None)
