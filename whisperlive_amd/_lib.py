"""ctypes binding of libwlx.so (include/wlx.h). There is NO CPU fallback: if the HIP library is missing or
fails to load, importing the engine raises and tells the user how to build it."""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
# WLX_LIB selects another build of the same sources (scripts/trace_step.py: libwlx_trace.so, compiled with -DWLX_TRACE)
DEFAULT_LIB = PKG_DIR / "libwlx.so"
LIB_PATH = Path(os.environ["WLX_LIB"]).resolve() if os.environ.get("WLX_LIB") else DEFAULT_LIB
SOURCES = ["pack.hip", "logmel.hip", "gemm.hip", "attention.hip", "decoder.hip", "search.hip", "engine.hip", "vad.hip"]
EXPORTS = [
    "wlx_abi_version", "wlx_last_error", "wlx_engine_create", "wlx_engine_destroy", "wlx_engine_spec",
    "wlx_slot_create", "wlx_slot_destroy", "wlx_logmel", "wlx_pcm_put", "wlx_logmel_resident", "wlx_features_get", "wlx_features_set", "wlx_encode",
    "wlx_encoder_output_get", "wlx_generate", "wlx_generate_ex", "wlx_detect_language", "wlx_align", "wlx_timings_get", "wlx_sync",
    "wlx_vad_create", "wlx_vad_destroy", "wlx_vad_probs",
    "wlx_ring_create", "wlx_ring_destroy", "wlx_ring_append", "wlx_ring_state", "wlx_vad_probs_resident", "wlx_vad_segments", "wlx_logmel_ring",
    "wlx_debug_logits_get", "wlx_debug_decode_logits", "wlx_debug_search", "wlx_debug_time_decode_step", "wlx_debug_profile_step", "wlx_debug_trace_step",
]


# MFMA accumulators in VGPRs (gfx950 has ONE register file; hipcc's default puts C/D in its AGPR half): every VALU touch of
# an accumulator — the softmax rescale of the attention kernel, the GEMM epilogues — otherwise costs a v_accvgpr_read and
# a v_accvgpr_write per register: a third of all VALU instructions of the encoder attention kernel (576 of 1738 per six key
# tiles), which is VALU-issue bound (rocprofv3: SQ_ACTIVE_INST_VALU 56 % of its wave cycles). No kernel spills with it.
HIPCC_EXTRA = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]


class WlxError(RuntimeError):
    pass


class wlx_spec(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_mels", "d_model", "n_heads", "enc_layers", "dec_layers", "ffn", "vocab", "n_audio_ctx", "n_text_ctx")]


class wlx_tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("on_device", C.c_int32)]


class wlx_token_ids(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("sot", "eot", "no_timestamps", "timestamp_begin", "no_speech", "blank")]


class wlx_gen_opts(C.Structure):
    _fields_ = [
        ("beam_size", C.c_int32), ("patience", C.c_float), ("num_hypotheses", C.c_int32),
        ("length_penalty", C.c_float), ("repetition_penalty", C.c_float), ("no_repeat_ngram_size", C.c_int32),
        ("max_length", C.c_int32), ("suppress_blank", C.c_int32), ("suppress_tokens", C.POINTER(C.c_int32)),
        ("n_suppress_tokens", C.c_int32), ("max_initial_timestamp_index", C.c_int32), ("sampling_topk", C.c_int32),
        ("sampling_temperature", C.c_float), ("seed", C.c_uint64), ("ids", wlx_token_ids),
    ]


class wlx_timings(C.Structure):
    _fields_ = [("logmel_ms", C.c_float), ("encode_ms", C.c_float), ("generate_ms", C.c_float), ("decode_steps", C.c_int32)]


class wlx_kernel_stat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches_per_step", C.c_float), ("avg_us", C.c_float),
                ("total_us_per_step", C.c_float), ("bytes_per_launch", C.c_double)]


class wlx_vad_weights(C.Structure):
    _fields_ = [("stft_basis", C.POINTER(C.c_float)), ("enc_w", C.POINTER(C.c_float) * 4), ("enc_b", C.POINTER(C.c_float) * 4),
                ("lstm_w_ih", C.POINTER(C.c_float)), ("lstm_w_hh", C.POINTER(C.c_float)), ("lstm_b_ih", C.POINTER(C.c_float)),
                ("lstm_b_hh", C.POINTER(C.c_float)), ("out_w", C.POINTER(C.c_float)), ("out_b", C.POINTER(C.c_float))]


_extra_ok = None


def _hipcc() -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise WlxError("hipcc not found: cannot build libwlx.so (ROCm toolchain required)")
    return hipcc


def _compile(out: Path, defines=(), verbose: bool = False, force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 of every source into `out`: one object per source (compiled in parallel, cached under
    csrc/.obj/<variant>/ and rebuilt when the source or any header is newer), linked into a temporary file and renamed
    (concurrent builders never see a half-written library). HIPCC_EXTRA is a hidden LLVM option: a hipcc that does not know
    it fails with 'Unknown command line argument' — retry once without it (the AGPR-form build: correct, the encoder
    attention ~8 % slower) and remember the answer for the other builds of this process."""
    global _extra_ok
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    defines = list(defines)
    hdrs = list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "wlx.h"]
    hdr_time = max(h.stat().st_mtime for h in hdrs)
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + [f"-D{d}" for d in defines]

    def objects(extra):
        tag = "_".join(["base" if extra else "agpr"] + [d.replace("=", "-") for d in defines])
        odir = CSRC / ".obj" / tag
        odir.mkdir(parents=True, exist_ok=True)

        def one(src):
            obj = odir / (src + ".o")
            sp = CSRC / src
            if not force and obj.exists() and obj.stat().st_mtime >= max(sp.stat().st_mtime, hdr_time):
                return obj, None           # (force=True: a new hipcc / ROCm or changed flags must not link stale objects)
            tmp = obj.with_name(obj.name + f".tmp{os.getpid()}")
            proc = subprocess.run(base + extra + ["-c", str(sp), "-o", str(tmp)], capture_output=True, text=True)
            if verbose and (proc.stdout or proc.stderr):
                print(proc.stdout, proc.stderr)
            if proc.returncode != 0:
                if tmp.exists():
                    tmp.unlink()
                return None, proc
            os.replace(tmp, obj)
            return obj, None
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
            return list(ex.map(one, SOURCES))

    for extra in ([HIPCC_EXTRA, []] if _extra_ok is not False else [[]]):
        res = objects(extra)
        bad = [p for o, p in res if o is None]
        if not bad:
            if extra:
                _extra_ok = True
            tmp = out.with_name(out.name + f".tmp{os.getpid()}")
            proc = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp)] + [str(o) for o, _ in res],
                                  capture_output=True, text=True)
            if proc.returncode != 0:
                if tmp.exists():
                    tmp.unlink()
                raise WlxError(f"hipcc link failed ({proc.returncode}):\n{proc.stderr[-4000:]}")
            os.replace(tmp, out)
            return out
        err = bad[0].stderr
        if extra and ("Unknown command line argument" in err or "unknown argument" in err.lower()):
            _extra_ok = False
            print(f"[wlx build] this hipcc rejects {' '.join(HIPCC_EXTRA)}: building with MFMA accumulators in AGPR form")
            continue
        raise WlxError(f"hipcc failed ({bad[0].returncode}):\n{err[-4000:]}")
    raise WlxError("hipcc failed")


def _fresh(out: Path) -> bool:
    deps = [CSRC / s for s in SOURCES] + list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "wlx.h"]
    return out.exists() and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps)


def build_trace() -> Path:
    """libwlx_trace.so: the same sources with -DWLX_TRACE (in-kernel timeline marks, profiling only)."""
    out = PKG_DIR / "libwlx_trace.so"
    return out if _fresh(out) else _compile(out, ["WLX_TRACE"])


def build_ab() -> Path:
    """libwlx_ab.so: the same sources with -DWLX_AB — the A/B switches that survive in the source (csrc/common.h wlx_ab: WLX_GEMM3,
    WLX_ROWTILE, WLX_RT_F16_NTB2, WLX_NO_FUSED_CQ, WLX_PREFILL_JOINT, WLX_DECODE_V1) read the
    environment in THIS library only; the production libwlx.so compiles them out. Select with WLX_LIB=<path>."""
    out = PKG_DIR / "libwlx_ab.so"
    return out if _fresh(out) else _compile(out, ["WLX_AB"])


def build_variant(name: str, defines) -> Path:
    """A/B builds of the same sources with extra -D flags (e.g. libwlx_wfirst.so: -DWLX_X_FIRST=0, the decode GEMVs with
    their weight stream requested BEFORE the activations, the round-1 order); selected at run time with WLX_LIB=<path>."""
    out = PKG_DIR / name
    return out if _fresh(out) else _compile(out, list(defines))


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 into whisperlive_amd/libwlx.so (hipcc cross-compiles without a GPU)."""
    if not force and _fresh(DEFAULT_LIB):
        return DEFAULT_LIB
    return _compile(DEFAULT_LIB, (), verbose, force=force)


_lib = None


def load() -> C.CDLL:
    """Load libwlx.so and declare prototypes. Raises WlxError (never falls back) if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise WlxError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). whisperlive_amd has no CPU fallback.")
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as e:  # e.g. libamdhip64 missing
        raise WlxError(f"cannot load {LIB_PATH}: {e}") from e
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise WlxError(f"libwlx.so lacks symbols {missing}")
    i32, i64, f32p, i32p, vp = C.c_int32, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    lib.wlx_abi_version.restype = i32
    lib.wlx_last_error.restype = C.c_char_p
    lib.wlx_engine_create.argtypes = [C.POINTER(wlx_spec), C.POINTER(wlx_tensor), i32, i32, C.POINTER(vp)]
    lib.wlx_engine_destroy.argtypes = [vp]
    lib.wlx_engine_destroy.restype = None
    lib.wlx_engine_spec.argtypes = [vp, C.POINTER(wlx_spec)]
    lib.wlx_slot_create.argtypes = [vp, i32, i32, i32p]
    lib.wlx_slot_destroy.argtypes = [vp, i32]
    lib.wlx_logmel.argtypes = [vp, i32, i32, f32p, i64, i32p]
    lib.wlx_pcm_put.argtypes = [vp, i32, i32, f32p, i64]
    lib.wlx_logmel_resident.argtypes = [vp, i32, i32, i32p]
    lib.wlx_features_get.argtypes = [vp, i32, i32, f32p, i64, i32p]
    lib.wlx_features_set.argtypes = [vp, i32, i32, f32p, i32, i32]
    lib.wlx_encode.argtypes = [vp, i32, i32, i32p, i32p]
    lib.wlx_encoder_output_get.argtypes = [vp, i32, i32, f32p, i64]
    lib.wlx_generate.argtypes = [vp, i32, i32, i32p, i32p, i32, C.POINTER(wlx_gen_opts), i32p, i32, i32p, f32p, f32p]
    lib.wlx_generate_ex.argtypes = [vp, i32, i32, i32p, i32p, i32p, i32, C.POINTER(wlx_gen_opts), i32p, i32, i32p, f32p, f32p]
    lib.wlx_detect_language.argtypes = [vp, i32, i32, i32, i32p, i32, f32p]
    lib.wlx_align.argtypes = [vp, i32, i32, i32p, i32, i32, i32, i32, i32p, i32, i32, i32p, i32p, i32, i32p, f32p]
    lib.wlx_timings_get.argtypes = [vp, i32, C.POINTER(wlx_timings)]
    lib.wlx_sync.argtypes = [vp, i32]
    lib.wlx_vad_create.argtypes = [C.POINTER(wlx_vad_weights), i32, C.POINTER(vp)]
    lib.wlx_vad_destroy.argtypes = [vp]
    lib.wlx_vad_destroy.restype = None
    lib.wlx_vad_probs.argtypes = [vp, f32p, i64, f32p, i32, i32p, f32p]
    i64p = C.POINTER(C.c_int64)
    lib.wlx_ring_create.argtypes = [vp, i64, C.POINTER(vp)]
    lib.wlx_ring_destroy.argtypes = [vp]
    lib.wlx_ring_destroy.restype = None
    lib.wlx_ring_append.argtypes = [vp, f32p, i64, i64, i64, i64p, i64p, i64p]
    lib.wlx_ring_state.argtypes = [vp, i64p, i64p]
    lib.wlx_vad_probs_resident.argtypes = [vp, vp, i64, i64, i32, f32p, i32, i32p, f32p]
    lib.wlx_vad_segments.argtypes = [f32p, i32, i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i64p, i32, i32p]
    lib.wlx_logmel_ring.argtypes = [vp, i32, i32, vp, i64p, i32, i32p]
    lib.wlx_debug_logits_get.argtypes = [vp, i32, f32p, i32, i64]
    lib.wlx_debug_decode_logits.argtypes = [vp, i32, i32p, i32, f32p]
    lib.wlx_debug_search.argtypes = [vp, i32, f32p, i32, i32p, i32, C.POINTER(wlx_gen_opts), i32p, i32, i32p, f32p]
    lib.wlx_debug_time_decode_step.argtypes = [vp, i32, i32, i32, i32, f32p]
    lib.wlx_debug_profile_step.argtypes = [vp, i32, i32, i32, i32, C.POINTER(wlx_kernel_stat), i32, i32p]
    lib.wlx_debug_trace_step.argtypes = [vp, i32, i32, i32, i32, C.POINTER(C.c_uint64), i64, C.c_char_p, i32p]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("wlx_last_error", "wlx_engine_destroy", "wlx_vad_destroy", "wlx_ring_destroy"):
            fn.restype = i32
    if lib.wlx_abi_version() != 1:
        raise WlxError("libwlx.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().wlx_last_error()
        raise WlxError(f"libwlx error {rc}: {msg.decode() if msg else '?'}")
