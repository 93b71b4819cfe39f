"""Python face of the C-ABI engine (include/wlx.h): one ``HipWhisperEngine`` per GPU, ``Slot`` objects for
concurrent streams. numpy in / numpy out; all arithmetic happens in libwlx.so on the MI355X."""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import WlxError, check
from .specs import WhisperSpec


def _f32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


@dataclass
class TokenIds:
    sot: int
    eot: int
    no_timestamps: int
    timestamp_begin: int
    no_speech: int
    blank: int = -1


@dataclass
class GenerationResult:
    """Mirror of ctranslate2.models.WhisperGenerationResult as the reference reads it
    (whisper_live/transcriber/transcriber_faster_whisper.py:1409-1414; whisper_live/batch_inference.py:357-368)."""
    sequences_ids: List[List[int]]
    scores: List[float]
    no_speech_prob: float
    sequences: Optional[List[List[str]]] = None


class HipWhisperEngine:
    """Weights repacked for gfx950 + kernels; thread-safe for concurrent calls on distinct slots."""

    def __init__(self, spec: WhisperSpec, weights: Dict[str, "np.ndarray"], device: int = 0):
        self.lib = _lib.load()
        self.spec = spec
        self.device = device
        cs = _lib.wlx_spec(spec.n_mels, spec.d_model, spec.n_heads, spec.enc_layers, spec.dec_layers, spec.ffn,
                           spec.vocab, spec.n_audio_ctx, spec.n_text_ctx)
        keep = []
        arr = (_lib.wlx_tensor * len(weights))()
        for i, (name, t) in enumerate(weights.items()):
            on_dev = 0
            if hasattr(t, "data_ptr"):            # torch tensor (PyTorch-ROCm holds the weights)
                import torch
                t = t.detach().to(torch.float32).contiguous()
                on_dev = 1 if t.is_cuda else 0
                ptr, shape = t.data_ptr(), tuple(t.shape)
            else:
                t = np.ascontiguousarray(t, dtype=np.float32)
                ptr, shape = t.ctypes.data, t.shape
            keep.append(t)
            arr[i].name = name.encode()
            arr[i].data = ptr
            arr[i].ndim = len(shape)
            for j, s in enumerate(shape):
                arr[i].shape[j] = s
            arr[i].on_device = on_dev
        h = C.c_void_p()
        check(self.lib.wlx_engine_create(C.byref(cs), arr, len(weights), device, C.byref(h)))
        self._h = h
        self._lock = threading.Lock()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wlx_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        import sys
        if sys.is_finalizing():      # interpreter shutdown: daemon session threads may still be inside a call; the
            return                   # process is about to return the device memory anyway
        try:
            self.close()
        except Exception:
            pass

    def create_ring(self, capacity_samples: int = 0) -> "PcmRing":
        """A device-resident PCM ring for one client stream (include/wlx.h wlx_ring_*)."""
        h = C.c_void_p()
        check(self.lib.wlx_ring_create(self._h, int(capacity_samples), C.byref(h)))
        return PcmRing(self, h)

    def create_slot(self, max_batch: int = 1, max_rows_per_item: int = 5) -> "Slot":
        sid = C.c_int32(-1)
        check(self.lib.wlx_slot_create(self._h, max_batch, max_rows_per_item, C.byref(sid)))
        return Slot(self, sid.value, max_batch, max_rows_per_item)


class PcmRing:
    """The device-side mirror of ServeClientBase.frames_np (whisper_live/backend/base.py:173-234): a client's packets are appended
    ONCE, the VAD gate and the log-mel front end read them in HBM. Positions are absolute stream sample positions."""
    MAX_RESIDENT = 45 * 16000          # base.py:191 (45 s) ...
    TRIM = 30 * 16000                  # ... :192-193 (the oldest 30 s go)

    def __init__(self, engine: "HipWhisperEngine", handle):
        self.engine, self.lib, self._h = engine, engine.lib, handle
        self.device = getattr(engine, "device", 0)

    def append(self, samples: np.ndarray, max_resident: Optional[int] = None, trim: Optional[int] = None) -> Tuple[int, int, int]:
        """-> (samples dropped by this call, first resident position, resident count)"""
        x = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
        d, b, r = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.wlx_ring_append(self._h, _f32p(x), x.shape[0], self.MAX_RESIDENT if max_resident is None else int(max_resident),
                                       self.TRIM if trim is None else int(trim), C.byref(d), C.byref(b), C.byref(r)))
        return d.value, b.value, r.value

    def state(self) -> Tuple[int, int]:
        b, r = C.c_int64(0), C.c_int64(0)
        check(self.lib.wlx_ring_state(self._h, C.byref(b), C.byref(r)))
        return b.value, r.value

    def close(self):
        if getattr(self, "_h", None):
            self.lib.wlx_ring_destroy(self._h)
            self._h = None


class Slot:
    """One unit of concurrency (own HIP stream + scratch). Not re-entrant: one call at a time per slot."""

    def __init__(self, engine: HipWhisperEngine, sid: int, max_batch: int, rows: int):
        self.engine, self.sid, self.max_batch, self.rows = engine, sid, max_batch, rows
        self.lib, self._h = engine.lib, engine._h
        self.lock = threading.Lock()

    def close(self):
        if self.sid >= 0 and self.engine._h:
            self.lib.wlx_slot_destroy(self.engine._h, self.sid)
        self.sid = -1

    # ---- features
    def logmel(self, pcm: np.ndarray, item: int = 0) -> int:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        nf = C.c_int32(0)
        check(self.lib.wlx_logmel(self.engine._h, self.sid, item, _f32p(pcm), pcm.shape[0], C.byref(nf)))
        return nf.value

    def pcm_put(self, pcm: np.ndarray, item: int = 0):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        check(self.lib.wlx_pcm_put(self.engine._h, self.sid, item, _f32p(pcm), pcm.shape[0]))

    def logmel_ring(self, ring: "PcmRing", ranges: Sequence[Tuple[int, int]], item: int = 0) -> int:
        """log-mel of the concatenation of ring ranges [(start, end), ...] (absolute positions) -> frames; see wlx_logmel_ring"""
        rg = np.asarray(ranges, dtype=np.int64).reshape(-1, 2)
        nf = C.c_int32(0)
        check(self.lib.wlx_logmel_ring(self.engine._h, self.sid, item, ring._h, rg.ctypes.data_as(C.POINTER(C.c_int64)), rg.shape[0], C.byref(nf)))
        return nf.value

    def logmel_resident(self, item: int = 0) -> int:
        nf = C.c_int32(0)
        check(self.lib.wlx_logmel_resident(self.engine._h, self.sid, item, C.byref(nf)))
        return nf.value

    def features(self, item: int = 0) -> np.ndarray:
        nf = C.c_int32(0)
        check(self.lib.wlx_features_get(self.engine._h, self.sid, item, None, 0, C.byref(nf)))
        out = np.empty((self.engine.spec.n_mels, nf.value), dtype=np.float32)
        check(self.lib.wlx_features_get(self.engine._h, self.sid, item, _f32p(out), out.size, C.byref(nf)))
        return out

    def set_features(self, feats: np.ndarray, item: int = 0):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        check(self.lib.wlx_features_set(self.engine._h, self.sid, item, _f32p(feats), feats.shape[0], feats.shape[1]))

    # ---- encoder
    def encode(self, batch: int = 1, seek: Optional[Sequence[int]] = None, seg: Optional[Sequence[int]] = None):
        sk = np.asarray(seek if seek is not None else [0] * batch, dtype=np.int32)
        if seg is None:
            sg_p = None
        else:
            sg = np.asarray(seg, dtype=np.int32)
            sg_p = _i32p(sg)
        check(self.lib.wlx_encode(self.engine._h, self.sid, batch, _i32p(sk), sg_p))

    def encoder_output(self, item: int = 0) -> np.ndarray:
        out = np.empty((self.engine.spec.n_audio_ctx, self.engine.spec.d_model), dtype=np.float32)
        check(self.lib.wlx_encoder_output_get(self.engine._h, self.sid, item, _f32p(out), out.size))
        return out

    # ---- decoder
    def _opts(self, ids: TokenIds, beam_size, patience, num_hypotheses, length_penalty, repetition_penalty,
              no_repeat_ngram_size, max_length, suppress_blank, suppress_tokens, max_initial_timestamp_index,
              sampling_topk, sampling_temperature, seed):
        o = _lib.wlx_gen_opts()
        o.beam_size, o.patience, o.num_hypotheses = int(beam_size), float(patience), int(num_hypotheses)
        o.length_penalty, o.repetition_penalty = float(length_penalty), float(repetition_penalty)
        o.no_repeat_ngram_size, o.max_length = int(no_repeat_ngram_size), int(max_length)
        o.suppress_blank = 1 if suppress_blank else 0
        st = np.asarray(list(suppress_tokens) if suppress_tokens is not None else [], dtype=np.int32)
        o.suppress_tokens = _i32p(st) if st.size else None
        o.n_suppress_tokens = int(st.size)
        o.max_initial_timestamp_index = int(max_initial_timestamp_index)
        o.sampling_topk, o.sampling_temperature, o.seed = int(sampling_topk), float(sampling_temperature), int(seed)
        o.ids = _lib.wlx_token_ids(ids.sot, ids.eot, ids.no_timestamps, ids.timestamp_begin, ids.no_speech, ids.blank)
        return o, st

    def generate(self, prompts: Sequence[Sequence[int]], ids: TokenIds, *, beam_size=5, patience=1.0, num_hypotheses=1,
                 length_penalty=1.0, repetition_penalty=1.0, no_repeat_ngram_size=0, max_length=448,
                 suppress_blank=True, suppress_tokens=(), max_initial_timestamp_index=50, sampling_topk=0,
                 sampling_temperature=0.0, seed=0, enc_items: Optional[Sequence[int]] = None) -> List[GenerationResult]:
        batch = len(prompts)
        o, keep = self._opts(ids, beam_size, patience, num_hypotheses, length_penalty, repetition_penalty,
                             no_repeat_ngram_size, max_length, suppress_blank, suppress_tokens,
                             max_initial_timestamp_index, sampling_topk, sampling_temperature, seed)
        stride = max(len(p) for p in prompts)
        pr = np.zeros((batch, stride), dtype=np.int32)
        pl = np.zeros(batch, dtype=np.int32)
        for i, p in enumerate(prompts):
            pr[i, :len(p)] = p
            pl[i] = len(p)
        nh = max(1, int(num_hypotheses))
        toks = np.zeros((batch, nh, 448), dtype=np.int32)
        nt = np.zeros((batch, nh), dtype=np.int32)
        sc = np.zeros((batch, nh), dtype=np.float32)
        nsp = np.zeros(batch, dtype=np.float32)
        items = np.asarray(enc_items, dtype=np.int32) if enc_items is not None else None
        check(self.lib.wlx_generate_ex(self.engine._h, self.sid, batch, _i32p(items) if items is not None else None,
                                       _i32p(pr), _i32p(pl), stride, C.byref(o),
                                       _i32p(toks), 448, _i32p(nt), _f32p(sc), _f32p(nsp)))
        out = []
        for b in range(batch):
            seqs = [toks[b, h, :nt[b, h]].tolist() for h in range(nh) if np.isfinite(sc[b, h])]
            scores = [float(sc[b, h]) for h in range(nh) if np.isfinite(sc[b, h])]
            out.append(GenerationResult(seqs, scores, float(nsp[b])))
        return out

    def align(self, tokens: Sequence[int], n_sot: int, num_frames: int, heads: Sequence[Tuple[int, int]], eot: int,
              median_filter_width: int = 7, item: int = 0):
        """ctranslate2 Whisper.align for one encoded item: tokens = start_sequence + [no_timestamps] + text + [eot].
        Returns (text_indices, time_indices, text_token_probs)."""
        tk = np.ascontiguousarray(tokens, dtype=np.int32)
        hd = np.ascontiguousarray(np.asarray(heads, dtype=np.int32).reshape(-1, 2))
        cap = tk.size + 1500 + 8
        ti = np.zeros(cap, dtype=np.int32)
        fi = np.zeros(cap, dtype=np.int32)
        n_path = C.c_int32(0)
        probs = np.zeros(max(1, tk.size - n_sot - 2), dtype=np.float32)
        check(self.lib.wlx_align(self.engine._h, self.sid, item, _i32p(tk), tk.size, n_sot, int(num_frames), int(median_filter_width),
                                 _i32p(hd), hd.shape[0], int(eot), _i32p(ti), _i32p(fi), cap, C.byref(n_path), _f32p(probs)))
        return ti[: n_path.value].copy(), fi[: n_path.value].copy(), probs

    def detect_language(self, batch: int, sot: int, lang_ids: Sequence[int]) -> np.ndarray:
        li = np.asarray(lang_ids, dtype=np.int32)
        probs = np.zeros((batch, li.size), dtype=np.float32)
        check(self.lib.wlx_detect_language(self.engine._h, self.sid, batch, sot, _i32p(li), li.size, _f32p(probs)))
        return probs

    def timings(self) -> dict:
        t = _lib.wlx_timings()
        check(self.lib.wlx_timings_get(self.engine._h, self.sid, C.byref(t)))
        return {"logmel_ms": t.logmel_ms, "encode_ms": t.encode_ms, "generate_ms": t.generate_ms, "decode_steps": t.decode_steps}

    # ---- test hooks
    def debug_decode_logits(self, tokens: Sequence[int]) -> np.ndarray:
        tk = np.asarray(tokens, dtype=np.int32)
        out = np.empty((tk.size, self.engine.spec.vocab), dtype=np.float32)
        check(self.lib.wlx_debug_decode_logits(self.engine._h, self.sid, _i32p(tk), tk.size, _f32p(out)))
        return out

    def debug_logits(self, rows: int) -> np.ndarray:
        """the slot's logits buffer [rows, V] as the LAST decoder pass left it (wlx_debug_logits_get)"""
        out = np.empty((rows, self.engine.spec.vocab), dtype=np.float32)
        check(self.lib.wlx_debug_logits_get(self.engine._h, self.sid, _f32p(out), rows, out.size))
        return out

    def debug_search(self, logits: np.ndarray, prompt: Sequence[int], ids: TokenIds, **kw) -> GenerationResult:
        logits = np.ascontiguousarray(logits, dtype=np.float32)   # [steps, rows, V]
        d = dict(beam_size=5, patience=1.0, num_hypotheses=1, length_penalty=1.0, repetition_penalty=1.0,
                 no_repeat_ngram_size=0, max_length=448, suppress_blank=True, suppress_tokens=(),
                 max_initial_timestamp_index=50, sampling_topk=0, sampling_temperature=0.0, seed=0)
        d.update(kw)
        o, keep = self._opts(ids, **d)
        pr = np.asarray(prompt, dtype=np.int32)
        nh = max(1, int(d["num_hypotheses"]))
        toks = np.zeros((nh, 448), dtype=np.int32)
        nt = np.zeros(nh, dtype=np.int32)
        sc = np.zeros(nh, dtype=np.float32)
        check(self.lib.wlx_debug_search(self.engine._h, self.sid, _f32p(logits), logits.shape[0], _i32p(pr), pr.size,
                                        C.byref(o), _i32p(toks), 448, _i32p(nt), _f32p(sc)))
        seqs = [toks[h, :nt[h]].tolist() for h in range(nh) if np.isfinite(sc[h])]
        return GenerationResult(seqs, [float(x) for x in sc if np.isfinite(x)], 0.0)

    def debug_time_decode_step(self, rows: int, t: int, iters: int = 50) -> float:
        ms = C.c_float(0)
        check(self.lib.wlx_debug_time_decode_step(self.engine._h, self.sid, rows, t, iters, C.byref(ms)))
        return ms.value

    def debug_profile_step(self, rows: int, t: int, iters: int = 20) -> List[dict]:
        cap = 64
        arr = (_lib.wlx_kernel_stat * cap)()
        n = C.c_int32(0)
        check(self.lib.wlx_debug_profile_step(self.engine._h, self.sid, rows, t, iters, arr, cap, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches_per_step, avg_us=arr[i].avg_us,
                     total_us=arr[i].total_us_per_step, bytes_per_launch=arr[i].bytes_per_launch) for i in range(n.value)]

    def debug_trace_step(self, rows: int, t: int, with_search: bool = True):
        """In-kernel timeline of one decode step (libwlx_trace.so only). Returns (names, records[n][2049][8] u64); record 0 of a launch is unused."""
        stride = (2048 + 1) * 8
        cap_launch = 320
        buf = np.zeros(cap_launch * stride, dtype=np.uint64)
        names = C.create_string_buffer(cap_launch * 48)
        n = C.c_int32(0)
        check(self.lib.wlx_debug_trace_step(self.engine._h, self.sid, rows, t, 1 if with_search else 0,
                                            buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size, names, C.byref(n)))
        nl = n.value
        nm = [names.raw[i * 48:(i + 1) * 48].split(b"\0")[0].decode() for i in range(nl)]
        return nm, buf[: nl * stride].reshape(nl, 2049, 8)
