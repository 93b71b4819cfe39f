"""CPU test: the C-ABI library builds for gfx950, loads, and exports every function include/wlx.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "wlx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wlx_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_documented_entry_points():
    fns = header_functions()
    for must in ("wlx_engine_create", "wlx_logmel", "wlx_encode", "wlx_generate", "wlx_detect_language", "wlx_last_error"):
        assert must in fns


def test_library_builds_loads_and_exports_every_declared_symbol():
    from whisperlive_amd import _lib
    path = _lib.build()
    assert os.path.isfile(path)
    lib = ctypes.CDLL(str(path))
    missing = [f for f in header_functions() if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == header_functions()        # python binding table == header
    lib.wlx_abi_version.restype = ctypes.c_int32
    assert lib.wlx_abi_version() == 1
    lib.wlx_last_error.restype = ctypes.c_char_p
    assert lib.wlx_last_error() is not None


def test_structs_match_header_sizes():
    from whisperlive_amd import _lib
    assert ctypes.sizeof(_lib.wlx_spec) == 9 * 4
    assert ctypes.sizeof(_lib.wlx_token_ids) == 6 * 4
    assert ctypes.sizeof(_lib.wlx_tensor) == 8 + 8 + 8 + 32 + 8          # name, data, ndim(+pad), shape[4], on_device(+pad)
    assert ctypes.sizeof(_lib.wlx_kernel_stat) == 64 + 3 * 4 + 4 + 8


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a GPU the engine must refuse to construct (no CPU fallback on the product path)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd._lib import WlxError
    from whisperlive_amd.specs import WhisperSpec
    from whisperlive_amd.weights import random_weights
    spec = WhisperSpec(80, 128, 2, 1, 1, 512, 2310)
    with pytest.raises(WlxError):
        HipWhisperEngine(spec, random_weights(spec, 0))


def test_vad_model_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oracle import silero_vad as sv
    from whisperlive_amd._lib import WlxError
    from whisperlive_amd.vad import SileroHIPModel
    with pytest.raises(WlxError):
        SileroHIPModel(sv.random_weights(0), device=0)
