"""WhisperModelHIP — the transcriber the reference's backend drives, on the MI355X engine.

Drop-in boundary #2 of SURVEY.md §8(b): the duck type of ``WhisperModel`` in
whisper_live/transcriber/transcriber_faster_whisper.py:574 as its two callers use it —
``ServeClientFasterWhisper.transcribe_audio`` (whisper_live/backend/faster_whisper_backend.py:236-244) and
``BatchInferenceWorker`` (whisper_live/batch_inference.py:257-258,271,283,294-308,346,355,393-402). Same method
names, argument meaning, return shapes and error behaviour; the five numerical call sites (feature extractor,
``model.encode``, ``model.generate``, ``model.detect_language``, ``StorageView.from_array``) go to libwlx.so through the
C-ABI (include/wlx.h) instead of faster-whisper / CTranslate2. The orchestration around them (seek loop, prompt
construction, temperature fallback, timestamp splitting, VAD time restoration) is host logic restated from
:692-1513 and :1792-1853 of that file. There is no CPU fallback: constructing the model without the HIP library or
without a GPU raises.

MI355X-first differences that do not change results: the PCM chunk is uploaded once and the log-mel features, the
30 s windows cut from them, the encoder output and the cross-attention K/V all stay resident in HBM between the
stages (the reference round-trips features through host numpy); one engine (= one copy of the weights) per GPU is
shared by every client, each client thread owning a *slot* (HIP stream + scratch) instead of a model replica.
"""
from __future__ import annotations

import itertools
import json
import logging
import os
import threading
import zlib
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import vad as _vad
from . import word_timing as _wt
from ._lib import WlxError as _WlxError
from .engine import GenerationResult, HipWhisperEngine, Slot, TokenIds
from .specs import WhisperSpec, get_spec, spec_from_state_dict
from .tokenizer import LANGUAGE_CODES, Tokenizer
from .types import Segment, TranscriptionInfo, TranscriptionOptions, Word  # noqa: F401
from .vad import VadOptions

logger = logging.getLogger("whisperlive_amd")


# ----------------------------------------------------------------------------------------------------------------
# small host helpers with the reference's semantics
def pad_or_trim(array: np.ndarray, length: int = 3000, axis: int = -1) -> np.ndarray:
    """Right-pad with ZEROS (in log-mel space) or trim (faster_whisper.audio.pad_or_trim; tensorrt_utils.py:80-104)."""
    n = array.shape[axis]
    if n > length:
        array = np.take(array, np.arange(length), axis=axis)
    elif n < length:
        widths = [(0, 0)] * array.ndim
        widths[axis] = (0, length - n)
        array = np.pad(array, widths)
    return array


def get_compression_ratio(text: str) -> float:
    raw = text.encode("utf-8")
    return len(raw) / len(zlib.compress(raw))


def get_suppressed_tokens(tokenizer: Tokenizer, suppress_tokens: Optional[Sequence[int]]) -> Tuple[int, ...]:
    """-1 expands to tokenizer.non_speech_tokens; the task / sot / sot_prev / sot_lm specials are always added
    (transcriber_faster_whisper.py:1831-1853)."""
    ids = [] if suppress_tokens is None else list(suppress_tokens)
    if -1 in ids:
        ids = [t for t in ids if t >= 0]
        ids.extend(tokenizer.non_speech_tokens)
    ids.extend([tokenizer.transcribe, tokenizer.translate, tokenizer.sot, tokenizer.sot_prev, tokenizer.sot_lm])
    return tuple(sorted(set(ids)))


def restore_speech_timestamps(segments: List[Segment], speech_chunks: List[dict], sampling_rate: int) -> List[Segment]:
    """Map times in the VAD-compressed audio back to the original timeline (:1792-1817)."""
    ts_map = _vad.SpeechTimestampsMap(speech_chunks, sampling_rate)
    for seg in segments:
        if seg.words:
            for w in seg.words:
                idx = ts_map.get_chunk_index((w.start + w.end) / 2)      # both ends resolved in the same chunk
                w.start = ts_map.get_original_time(w.start, idx)
                w.end = ts_map.get_original_time(w.end, idx)
            seg.start, seg.end = seg.words[0].start, seg.words[-1].end
        else:
            seg.start = ts_map.get_original_time(seg.start)
            seg.end = ts_map.get_original_time(seg.end)
    return segments


# ----------------------------------------------------------------------------------------------------------------
@dataclass
class DeviceFeatures:
    """Log-mel features living in a slot's HBM buffer: [n_mels, n_frames] of item `item` (n_frames incl. the pad frame)."""
    slot: Slot
    n_frames: int
    item: int = 0

    @property
    def shape(self):
        return (self.slot.engine.spec.n_mels, self.n_frames)

    def numpy(self) -> np.ndarray:
        return self.slot.features(self.item)


@dataclass
class ResidentAudio:
    """A chunk of a client's stream that is resident in a device PCM ring (whisperlive_amd.engine.PcmRing): samples
    [start, start + n) in absolute stream positions, plus the host copy the session holds anyway (whisper_live/backend/base.py:219-234
    hands `transcribe_audio` a numpy chunk) — the fallback when the range has been trimmed away meanwhile or the VAD model has no
    device path. `transcribe(ResidentAudio(...))` then moves NO audio over PCIe: the gate and the log-mel read the ring."""
    ring: "object"
    start: int
    n: int
    host: np.ndarray


@dataclass
class AlignmentResult:
    """shape of ctranslate2.models.WhisperAlignmentResult"""
    alignments: List[Tuple[int, int]]
    text_token_probs: List[float]


@dataclass
class EncoderOutput:
    """Encoder states + cross-attention K/V of `batch` items resident in a slot (what a CT2 StorageView stands for)."""
    slot: Slot
    batch: int
    generation: int
    items: Optional[List[int]] = None      # decoder item i -> encoder item items[i] (None = identity)

    @property
    def shape(self):
        sp = self.slot.engine.spec
        return (self.batch, sp.n_audio_ctx, sp.d_model)

    def numpy(self) -> np.ndarray:
        idx = self.items if self.items is not None else range(self.batch)
        return np.stack([self.slot.encoder_output(i) for i in idx])

    def select(self, indices: Sequence[int]) -> "EncoderOutput":
        """A sub-batch view over the same HBM-resident encoder output (no re-encode)."""
        base = self.items if self.items is not None else list(range(self.batch))
        return EncoderOutput(self.slot, len(indices), self.generation, [base[i] for i in indices])


class FeatureExtractorHIP:
    """Role of faster_whisper.feature_extractor.FeatureExtractor (attributes read at
    transcriber_faster_whisper.py:656-665,1058,1115-1126; called at :862 and batch_inference.py:258)."""

    def __init__(self, model: "WhisperModelHIP", feature_size: int = 80, sampling_rate: int = 16000, hop_length: int = 160,
                 chunk_length: int = 30, n_fft: int = 400):
        if (sampling_rate, hop_length, n_fft) != (16000, 160, 400):
            raise ValueError("the HIP log-mel kernel implements Whisper's 16 kHz / n_fft 400 / hop 160 front-end only")
        self._model = model
        self.n_mels = self.feature_size = feature_size
        self.sampling_rate, self.hop_length, self.n_fft, self.chunk_length = sampling_rate, hop_length, n_fft, chunk_length
        self.n_samples = chunk_length * sampling_rate
        self.nb_max_frames = self.n_samples // hop_length
        self.time_per_frame = hop_length / sampling_rate

    def __call__(self, waveform: np.ndarray, padding: int = 160, chunk_length: Optional[int] = None) -> np.ndarray:
        if padding != 160:
            raise ValueError("padding must be 160 (the value the reference uses)")
        slot = self._model._slot()
        with slot.lock:
            slot.logmel(np.asarray(waveform, dtype=np.float32))
            return slot.features()


class _EngineModel:
    """The ctranslate2.models.Whisper surface the reference calls."""

    MAX_DEC_ROWS = 320      # decoder rows one step covers (csrc/decoder.h WLX_MAX_DEC_ROWS: 16-row tiles; 64 until round 5, 48 until round 4)

    def __init__(self, owner: "WhisperModelHIP"):
        self._o = owner
        self.device = "cuda"
        self.device_index = [owner.engine.device]

    @property
    def is_multilingual(self) -> bool:
        return self._o._multilingual

    @property
    def n_mels(self) -> int:
        return self._o.spec.n_mels

    def encode(self, features, to_cpu: bool = False) -> EncoderOutput:
        return self._o.encode(features)

    def generate(self, encoder_output: EncoderOutput, prompts: Sequence[Sequence[int]], *, beam_size: int = 5,
                 patience: float = 1, num_hypotheses: int = 1, length_penalty: float = 1, repetition_penalty: float = 1,
                 no_repeat_ngram_size: int = 0, max_length: int = 448, return_scores: bool = True,
                 return_no_speech_prob: bool = True, max_initial_timestamp_index: int = 50, suppress_blank: bool = True,
                 suppress_tokens: Optional[Sequence[int]] = (-1,), sampling_topk: int = 1,
                 sampling_temperature: float = 1.0) -> List[GenerationResult]:
        o = self._o
        if not isinstance(encoder_output, EncoderOutput):
            encoder_output = o.encode(encoder_output)       # CT2 also accepts raw features here
        slot = encoder_output.slot
        if encoder_output.generation != slot._enc_generation:
            raise RuntimeError("stale encoder output: the slot has encoded another batch since")
        if len(prompts) != encoder_output.batch:
            raise ValueError(f"{len(prompts)} prompts for an encoder batch of {encoder_output.batch}")
        sup = list(suppress_tokens) if suppress_tokens is not None else []
        if -1 in sup:
            sup = [t for t in sup if t >= 0] + list(o._base_tokenizer.non_speech_tokens)
        # CT2's generate defaults: beam_size > 1 -> beam search; beam_size == 1 -> sampling with top-k / temperature.
        temp = 0.0 if beam_size > 1 else (float(sampling_temperature) if sampling_topk != 1 else 0.0)
        # One decode step covers every row the slot holds (max_batch x rows per item <= 320 since round 5; it was 64 rows — 12 clips at
        # beam 5 — and wider batches were cut into groups). A call that brings more rows than the slot (a duck-typed slot, a decode with
        # fewer rows per item than the slot was sized for) is still decoded as consecutive groups over the SAME resident encoder output
        # (item maps): identical results.
        rows_per_item = max(1, beam_size if (beam_size > 1 and temp == 0.0) else num_hypotheses)
        slot_rows = getattr(slot, "rows", None)
        if isinstance(slot_rows, int) and rows_per_item > slot_rows:
            raise ValueError(f"this transcriber's slots hold {slot_rows} decoder rows per audio item (max_batch={o.max_batch}), "
                             f"the decode asks for {rows_per_item} (beam_size / num_hypotheses)")
        slot_items = getattr(slot, "max_batch", None)
        cap_rows = slot_rows * slot_items if isinstance(slot_rows, int) and isinstance(slot_items, int) else 64
        group = max(1, min(self.MAX_DEC_ROWS, max(cap_rows, rows_per_item)) // rows_per_item)
        items = encoder_output.items if encoder_output.items is not None else list(range(encoder_output.batch))
        res = []
        for a in range(0, len(prompts), group):
            res.extend(slot.generate(prompts[a:a + group], o.token_ids, beam_size=beam_size, patience=patience,
                                     num_hypotheses=num_hypotheses, length_penalty=length_penalty,
                                     repetition_penalty=repetition_penalty, no_repeat_ngram_size=no_repeat_ngram_size,
                                     max_length=max_length, suppress_blank=suppress_blank, suppress_tokens=sup,
                                     max_initial_timestamp_index=max_initial_timestamp_index, sampling_topk=sampling_topk,
                                     sampling_temperature=temp, seed=o._next_seed(),
                                     enc_items=(items[a:a + group] if (encoder_output.items is not None or len(prompts) > group) else None)))
        hf = o.hf_tokenizer
        for r in res:
            tb = o.token_ids.timestamp_begin       # ids past the tokenizer's own table are timestamps (see __init__)
            r.sequences = [[hf.id_to_token(t) or f"<|{(t - tb) * 0.02:.2f}|>" for t in seq] for seq in r.sequences_ids]
        return res

    def align(self, encoder_output: EncoderOutput, start_sequence: Sequence[int], text_tokens: Sequence[Sequence[int]],
              num_frames: Union[int, Sequence[int]], *, median_filter_width: int = 7) -> List["AlignmentResult"]:
        """ctranslate2 Whisper.align (:1657-1663): one result per item with .alignments [(text index, time index)] and
        .text_token_probs"""
        o = self._o
        slot = encoder_output.slot
        if encoder_output.generation != slot._enc_generation:
            raise RuntimeError("stale encoder output: the slot has encoded another batch since")
        if len(text_tokens) != encoder_output.batch:
            raise ValueError(f"{len(text_tokens)} token lists for an encoder batch of {encoder_output.batch}")
        frames = [num_frames] * len(text_tokens) if isinstance(num_frames, int) else list(num_frames)
        bt = o._base_tokenizer
        out = []
        for i, toks in enumerate(text_tokens):
            seq = list(start_sequence) + [bt.no_timestamps] + list(toks) + [bt.eot]
            ti, fi, probs = slot.align(seq, len(start_sequence), frames[i], o.alignment_heads, bt.eot,
                                       median_filter_width=median_filter_width,
                                       item=encoder_output.items[i] if encoder_output.items is not None else i)
            out.append(AlignmentResult(list(zip(ti.tolist(), fi.tolist())), probs.tolist()))
        return out

    def detect_language(self, encoder_output: EncoderOutput) -> List[List[Tuple[str, float]]]:
        o = self._o
        if not isinstance(encoder_output, EncoderOutput):
            encoder_output = o.encode(encoder_output)
        if not self.is_multilingual:
            raise RuntimeError("detect_language on an English-only model")
        langs = o._base_tokenizer.language_token_ids()
        if encoder_output.items is not None:
            raise ValueError("detect_language needs the full encoder batch, not a selected view")
        probs = encoder_output.slot.detect_language(encoder_output.batch, o.token_ids.sot, [i for _, i in langs])
        out = []
        for b in range(encoder_output.batch):
            pairs = [(f"<|{c}|>", float(p)) for (c, _), p in zip(langs, probs[b])]
            out.append(sorted(pairs, key=lambda x: -x[1]))
        return out


class WhisperModelHIP:
    """See module docstring. `model_size_or_path`: a CTranslate2 / Hugging Face Whisper checkpoint directory (model.bin or
    model.safetensors + tokenizer.json [+ preprocessor_config.json]), a size name or hub id resolved like the reference's
    (whisperlive_amd/artifacts.py: cache first, download where allowed) or, with `weights=`/`hf_tokenizer=`, just an identifier."""

    def __init__(self, model_size_or_path: str = "small.en", device: str = "cuda", device_index: int = 0,
                 compute_type: str = "float16", *, weights: Optional[Dict[str, np.ndarray]] = None,
                 spec: Optional[WhisperSpec] = None, hf_tokenizer=None, engine: Optional[HipWhisperEngine] = None,
                 max_batch: int = 1, multilingual: Optional[bool] = None, vad_model=None, **_ignored):
        if device not in ("cuda", "auto"):
            raise ValueError("WhisperModelHIP runs on the MI355X only (device='cuda'); there is no CPU path")
        if compute_type not in ("float16", "default", "auto"):
            raise ValueError("compute_type: the HIP engine computes in float16 MFMA with float32 accumulation")
        self.logger = logger
        feat_kwargs: dict = {}
        if engine is not None:
            self.engine, self.spec = engine, engine.spec
        else:
            if weights is None:
                if not os.path.isdir(model_size_or_path):
                    # a size name or hub id, as the reference accepts them (transcriber_faster_whisper.py:620-632,
                    # faster_whisper_backend.py:133-178): cache first, download where allowed — raises FileNotFoundError
                    # (ArtifactNotFound) with every place looked at
                    from .artifacts import resolve_model
                    model_size_or_path = resolve_model(model_size_or_path, download_root=_ignored.get("download_root"),
                                                       local_files_only=_ignored.get("local_files_only"))
                from .weights import load_model_dir
                weights = load_model_dir(model_size_or_path)      # CTranslate2 model.bin or Hugging Face safetensors
            self.spec = spec or spec_from_state_dict(weights)
            self.engine = HipWhisperEngine(self.spec, weights, device=device_index)
        if hf_tokenizer is None:
            tok_file = os.path.join(model_size_or_path, "tokenizer.json") if os.path.isdir(model_size_or_path) else None
            if not tok_file or not os.path.isfile(tok_file):
                raise FileNotFoundError("tokenizer.json not found: pass hf_tokenizer= (no hub access offline)")
            import tokenizers
            hf_tokenizer = tokenizers.Tokenizer.from_file(tok_file)
        if os.path.isdir(model_size_or_path):
            cfg = os.path.join(model_size_or_path, "preprocessor_config.json")
            if os.path.isfile(cfg):
                try:
                    with open(cfg, encoding="utf-8") as f:
                        raw = json.load(f)
                    feat_kwargs = {k: raw[k] for k in ("feature_size", "sampling_rate", "hop_length", "chunk_length", "n_fft") if k in raw}
                except json.JSONDecodeError as e:        # same tolerance as :687-688
                    self.logger.warning("Could not load preprocessor config: %s", e)
        self.hf_tokenizer = hf_tokenizer
        self._multilingual = self.spec.multilingual if multilingual is None else bool(multilingual)
        # Older converted checkpoints ship a tokenizer.json WITHOUT the <|0.00|>..<|30.00|> entries (≈50364 ids for a
        # 51865-row model); faster-whisper derives timestamp ids as no_timestamps + 1 and never looks them up, so the
        # only hard requirements are: no id beyond the model's vocabulary, and every named special (and the 1501
        # timestamp ids derived from no_timestamps) inside it.
        if hf_tokenizer.get_vocab_size() > self.spec.vocab:
            raise ValueError(f"tokenizer has {hf_tokenizer.get_vocab_size()} ids, the model only {self.spec.vocab}")
        self._base_tokenizer = Tokenizer(hf_tokenizer, False)
        bt = self._base_tokenizer
        self.token_ids = TokenIds(bt.sot, bt.eot, bt.no_timestamps, bt.timestamp_begin, bt.no_speech, bt.blank)
        specials = dict(sot=bt.sot, eot=bt.eot, no_timestamps=bt.no_timestamps, no_speech=bt.no_speech,
                        transcribe=bt.transcribe, translate=bt.translate, sot_prev=bt.sot_prev, sot_lm=bt.sot_lm)
        bad = {k: v for k, v in specials.items() if not 0 <= v < self.spec.vocab}
        if bad or bt.timestamp_begin + 1500 > self.spec.vocab - 1:
            raise ValueError(f"tokenizer specials do not fit the model's {self.spec.vocab}-row vocabulary: {bad or 'timestamps'}")
        feat_kwargs.setdefault("feature_size", self.spec.n_mels)
        if feat_kwargs["feature_size"] != self.spec.n_mels:
            raise ValueError("preprocessor feature_size does not match the model's n_mels")
        self.feat_kwargs = feat_kwargs
        self.feature_extractor = FeatureExtractorHIP(self, **feat_kwargs)
        self.model = _EngineModel(self)
        # alignment heads for word timestamps: the checkpoint's own list (generation_config.json / config.json
        # "alignment_heads", as CT2's converter stores it) or the published default (upper half of the decoder)
        self.alignment_heads = _wt.default_alignment_heads(self.spec.dec_layers, self.spec.n_heads)
        if os.path.isdir(model_size_or_path):
            for name in ("generation_config.json", "config.json"):
                fp = os.path.join(model_size_or_path, name)
                try:
                    with open(fp, encoding="utf-8") as f:
                        heads = json.load(f).get("alignment_heads")
                    if heads:
                        self.alignment_heads = [(int(l), int(h)) for l, h in heads]
                        break
                except (OSError, ValueError, TypeError):
                    continue
        self.input_stride = 2
        self.num_samples_per_token = self.feature_extractor.hop_length * self.input_stride
        self.frames_per_second = self.feature_extractor.sampling_rate // self.feature_extractor.hop_length
        self.tokens_per_second = self.feature_extractor.sampling_rate // self.num_samples_per_token
        self.time_precision = 0.02
        self.max_length = 448
        self.max_batch = max_batch
        self.vad_model = vad_model          # padded audio -> per-window speech probability; None = the process's Silero weights on THIS GPU
        self._tls = threading.local()
        self._slots: List[Slot] = []
        self._slots_lock = threading.Lock()
        self._seed = itertools.count(0x5EED)

    def _vad_model(self):
        """the gate's probability model: the one given at construction, else the process's Silero weights on this
        transcriber's own GPU (raises vad.VadUnavailable when none are configured)"""
        return self.vad_model if self.vad_model is not None else _vad.get_default_model(getattr(self.engine, "device", 0))

    # ---- slots: one per calling thread (the reference runs one transcription thread per client)
    def _slot(self, rows: int = 5) -> Slot:
        """The calling thread's slot. A slot is a few hundred MB of HBM (encoder activations, cross K/V, KV cache), so
        slots are POOLED: a thread that needs one first takes over a slot whose owner thread has ended (a client that
        disconnected) or that was handed back with release_slot(); a new one is created only when every slot is in use.
        The pool therefore holds max-concurrent-clients slots, not one per connection ever made, and reconnecting
        clients reuse the captured decode graphs of their predecessor.
        `rows`: decoder rows per audio item the caller is about to need (beam_size / best_of; 5 = the reference's defaults). A call
        that asks for more than the thread's slot holds (the reference passes any beam_size through to CTranslate2) gets a wider slot
        (up to the engine's 16 rows per item); the narrower one is closed."""
        rows = max(5, int(rows))
        # validate BEFORE anything is closed: a refused request must leave the thread's slot — resident encoder output,
        # captured graphs, any EncoderOutput / DeviceFeatures the caller still holds — untouched
        if rows > 16:
            raise ValueError(f"beam_size / best_of {rows}: the engine decodes at most 16 rows per audio item")
        s = getattr(self._tls, "slot", None)
        old = None
        if s is not None and s.sid >= 0 and isinstance(getattr(s, "rows", None), int) and s.rows < rows:
            old, s = s, None                    # a wider slot is needed: the narrower one is closed once its replacement exists
        if s is None or s.sid < 0:
            me = threading.current_thread()
            with self._slots_lock:
                self._slots = [x for x in self._slots if x.sid >= 0]
                s = next((x for x in self._slots if x is not old and (x._owner is None or not x._owner.is_alive()) and getattr(x, "rows", rows) >= rows), None)
                if s is not None:
                    s._owner = me
            if s is None:
                # (5 rows per item = the reference's beam_size / best_of; the engine takes up to 64 items x 5 rows = 320 rows per slot)
                if self.max_batch * rows > 320:
                    raise ValueError(f"max_batch {self.max_batch} x {rows} rows per item = {self.max_batch * rows} decoder rows: a slot holds at "
                                     f"most 320 (64 items x 5 rows); lower beam_size / best_of or max_batch")
                s = self.engine.create_slot(self.max_batch, rows)        # raises: `old` stays the thread's slot
                s._enc_generation = 0
                s._owner = me
                with self._slots_lock:
                    self._slots.append(s)
            self._tls.slot = s
            if old is not None:
                with self._slots_lock:
                    self._slots = [x for x in self._slots if x is not old]
                old.close()
        return s

    def release_slot(self):
        """Hand the calling thread's slot back to the pool (a session thread calls this when it exits)."""
        s = getattr(self._tls, "slot", None)
        if s is not None:
            self._tls.slot = None
            with self._slots_lock:
                s._owner = None

    def _next_seed(self) -> int:
        return next(self._seed)

    def close(self):
        with self._slots_lock:
            for s in self._slots:
                s.close()
            self._slots.clear()

    @property
    def supported_languages(self) -> List[str]:
        return list(LANGUAGE_CODES) if self.model.is_multilingual else ["en"]

    # ---- encoder (transcriber_faster_whisper.py:1339-1348)
    def encode(self, features: Union[np.ndarray, DeviceFeatures], seek: int = 0, segment_size: Optional[int] = None
               ) -> EncoderOutput:
        slot = self._slot()
        if isinstance(features, DeviceFeatures):
            seg = features.n_frames - seek if segment_size is None else segment_size
            features.slot.encode(1, seek=[seek], seg=[seg])
            slot = features.slot
            batch = 1
        else:
            f = np.asarray(features, dtype=np.float32)
            if f.ndim == 2:
                f = f[None]
            if f.ndim != 3 or f.shape[1] != self.spec.n_mels:
                raise ValueError(f"features must be [batch, {self.spec.n_mels}, frames]")
            batch = f.shape[0]
            if batch > slot.max_batch:
                raise ValueError(f"batch {batch} exceeds the slot's max_batch {slot.max_batch}")
            for i in range(batch):
                slot.set_features(f[i], item=i)
            slot.encode(batch, seek=[0] * batch, seg=[min(f.shape[2], 3000)] * batch)
        slot._enc_generation += 1
        return EncoderOutput(slot, batch, slot._enc_generation)

    def encode_audio_batch(self, audios: Sequence[np.ndarray]) -> EncoderOutput:
        """Batched front half of whisper_live/batch_inference.py:236-271 on the device: per-item log-mel kernels
        into the slot's feature buffers, then ONE encoder launch chain over the batch. Keeps the reference's framing:
        feature_extractor(audio) -> pad_or_trim -> only the first 3000 frames (incl. the trailing pad frame) are used."""
        slot = self._slot()
        n = len(audios)
        if n > slot.max_batch:
            raise ValueError(f"batch {n} exceeds the slot's max_batch {slot.max_batch}")
        frames = [slot.logmel(np.ascontiguousarray(a, dtype=np.float32), item=i) for i, a in enumerate(audios)]
        slot.encode(n, seek=[0] * n, seg=[min(t, 3000) for t in frames])
        slot._enc_generation += 1
        return EncoderOutput(slot, n, slot._enc_generation)

    # ---- transcribe (transcriber_faster_whisper.py:692-968)
    def transcribe(self, audio: np.ndarray, language: Optional[str] = None, task: str = "transcribe",
                   log_progress: bool = False, beam_size: int = 5, best_of: int = 5, patience: float = 1,
                   length_penalty: float = 1, repetition_penalty: float = 1, no_repeat_ngram_size: int = 0,
                   temperature: Union[float, Sequence[float]] = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0),
                   compression_ratio_threshold: Optional[float] = 2.4, log_prob_threshold: Optional[float] = -1.0,
                   no_speech_threshold: Optional[float] = 0.6, condition_on_previous_text: bool = True,
                   prompt_reset_on_temperature: float = 0.5, initial_prompt: Optional[Union[str, Iterable[int]]] = None,
                   prefix: Optional[str] = None, suppress_blank: bool = True, suppress_tokens: Optional[List[int]] = (-1,),
                   without_timestamps: bool = False, max_initial_timestamp: float = 1.0, word_timestamps: bool = False,
                   prepend_punctuations: str = "\"'“¿([{-", append_punctuations: str = "\"'.。,，!！?？:：”)]}、",
                   multilingual: bool = False, vad_filter: bool = False,
                   vad_parameters: Optional[Union[dict, VadOptions]] = None, max_new_tokens: Optional[int] = None,
                   chunk_length: Optional[int] = None, clip_timestamps: Union[str, List[float]] = "0",
                   hallucination_silence_threshold: Optional[float] = None, hotwords: Optional[str] = None,
                   language_detection_threshold: Optional[float] = 0.5, language_detection_segments: int = 1,
                   ) -> Tuple[Optional[List[Segment]], Optional[TranscriptionInfo]]:
        sr = self.feature_extractor.sampling_rate
        if multilingual and not self.model.is_multilingual:
            self.logger.warning("The current model is English-only but the multilingual parameter is set to True; "
                                "setting to False instead.")
            multilingual = False
        resident = None
        if isinstance(audio, ResidentAudio):
            resident, audio = audio, audio.host
        if not isinstance(audio, np.ndarray):
            # path / bytes / file object, like the reference's decode_audio (:821); WAV and FLAC are read natively
            # (whisperlive_amd/audio_io.py), anything else must arrive as 16 kHz float32 PCM
            if not isinstance(audio, (str, bytes, bytearray)) and not hasattr(audio, "read"):
                raise TypeError("audio must be a float32 numpy waveform at 16 kHz, or a WAV / FLAC path, bytes or file object")
            from .audio_io import load_audio
            audio = load_audio(audio, sampling_rate=sr)
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        duration = audio.shape[0] / sr
        duration_after_vad = duration
        speech_chunks = None
        # the resident form of the same chunk (ResidentAudio): ring ranges the log-mel kernel will walk; None = the host path
        ranges = [(resident.start, resident.start + resident.n)] if resident is not None and resident.n == audio.shape[0] else None
        if vad_filter and clip_timestamps == "0":
            if vad_parameters is None:
                vad_parameters = VadOptions()
            elif isinstance(vad_parameters, dict):
                vad_parameters = VadOptions(**vad_parameters)
            vad_model = self._vad_model()
            if ranges is not None and hasattr(vad_model, "probs_resident") and getattr(vad_model, "device", None) == getattr(resident.ring, "device", -1):
                try:
                    speech_chunks = _vad.get_speech_timestamps_resident(resident.ring, resident.start, resident.n, vad_parameters, model=vad_model)
                except _WlxError as e:             # trimmed away under us (a backlog of > 45 s): the host copy is still whole
                    self.logger.debug("resident VAD fell back to the host chunk: %s", e)
                    ranges = None
            else:
                ranges = None
            if ranges is None:
                speech_chunks = _vad.get_speech_timestamps(audio, vad_parameters, model=vad_model)
                chunks, _meta = _vad.collect_chunks(audio, speech_chunks)
                audio = np.concatenate(chunks, axis=0)
                n_after = audio.shape[0]
            else:                                   # collect_chunks + np.concatenate, as ring ranges
                ranges = [(resident.start + c["start"], resident.start + c["end"]) for c in speech_chunks if c["end"] > c["start"]]
                n_after = sum(b - a for a, b in ranges)
                if len(ranges) > 256:               # (include/wlx.h wlx_logmel_ring takes <= 256 ranges)
                    chunks, _meta = _vad.collect_chunks(audio, speech_chunks)
                    audio, ranges = np.concatenate(chunks, axis=0), None
            duration_after_vad = n_after / sr
        else:
            n_after = audio.shape[0]
        if n_after == 0:
            return None, None                                 # reference patch, :860-861
        temps_ = temperature if isinstance(temperature, (list, tuple)) else [temperature]
        slot = self._slot(rows=max(int(beam_size), int(best_of) if any(t > 0 for t in temps_) else 1))
        with slot.lock:
            n_frames = None
            if ranges is not None:
                try:
                    n_frames = slot.logmel_ring(resident.ring, ranges)      # PCM already in HBM: the kernel walks the speech ranges
                except _WlxError as e:
                    self.logger.debug("resident log-mel fell back to the host chunk: %s", e)
                    if speech_chunks is not None:
                        chunks, _meta = _vad.collect_chunks(audio, speech_chunks)
                        audio = np.concatenate(chunks, axis=0)
            if n_frames is None:
                n_frames = slot.logmel(audio)                  # PCM -> HBM -> log-mel, stays on the device
            features = DeviceFeatures(slot, n_frames)
            all_language_probs = None
            if language is None:
                if not self.model.is_multilingual:
                    language, language_probability = "en", 1
                else:
                    start_ts = float(clip_timestamps.split(",")[0]) if isinstance(clip_timestamps, str) else clip_timestamps[0]
                    content_frames = n_frames - 1
                    seek = int(start_ts * self.frames_per_second) if start_ts * self.frames_per_second < content_frames else 0
                    language, language_probability, all_language_probs = self.detect_language(
                        features=features, language_detection_segments=language_detection_segments,
                        language_detection_threshold=language_detection_threshold, _seek=seek)
            else:
                if not self.model.is_multilingual and language != "en":
                    self.logger.warning("The current model is English-only but the language parameter is set to '%s'; "
                                        "using 'en' instead." % language)
                    language = "en"
                language_probability = 1
            tokenizer = Tokenizer(self.hf_tokenizer, self.model.is_multilingual, task=task, language=language)
            options = TranscriptionOptions(
                beam_size=beam_size, best_of=best_of, patience=patience, length_penalty=length_penalty,
                repetition_penalty=repetition_penalty, no_repeat_ngram_size=no_repeat_ngram_size,
                log_prob_threshold=log_prob_threshold, no_speech_threshold=no_speech_threshold,
                compression_ratio_threshold=compression_ratio_threshold,
                condition_on_previous_text=condition_on_previous_text,
                prompt_reset_on_temperature=prompt_reset_on_temperature,
                temperatures=list(temperature) if isinstance(temperature, (list, tuple)) else [temperature],
                initial_prompt=initial_prompt, prefix=prefix, suppress_blank=suppress_blank,
                suppress_tokens=get_suppressed_tokens(tokenizer, suppress_tokens) if suppress_tokens else suppress_tokens,
                without_timestamps=without_timestamps, max_initial_timestamp=max_initial_timestamp,
                word_timestamps=word_timestamps, prepend_punctuations=prepend_punctuations,
                append_punctuations=append_punctuations, multilingual=multilingual, max_new_tokens=max_new_tokens,
                clip_timestamps=clip_timestamps, hallucination_silence_threshold=hallucination_silence_threshold,
                hotwords=hotwords)
            segments = self.generate_segments(features, tokenizer, options, log_progress, None)
        if speech_chunks:
            segments = restore_speech_timestamps(segments, speech_chunks, sr)
        info = TranscriptionInfo(language=language, language_probability=language_probability, duration=duration,
                                 duration_after_vad=duration_after_vad, transcription_options=options,
                                 vad_options=vad_parameters, all_language_probs=all_language_probs)
        return segments, info

    # ---- timestamp splitting (:970-1047)
    def _split_segments_by_timestamps(self, tokenizer: Tokenizer, tokens: List[int], time_offset: float,
                                      segment_size: int, segment_duration: float, seek: int):
        tb = tokenizer.timestamp_begin
        out = []
        ends_on_single_ts = len(tokens) >= 2 and tokens[-2] < tb <= tokens[-1]
        pair_ends = [i for i in range(1, len(tokens)) if tokens[i] >= tb and tokens[i - 1] >= tb]
        if pair_ends:
            cuts = list(pair_ends)
            if ends_on_single_ts:
                cuts.append(len(tokens))
            last = 0
            for cut in cuts:
                piece = tokens[last:cut]
                out.append(dict(seek=seek, start=time_offset + (piece[0] - tb) * self.time_precision,
                                end=time_offset + (piece[-1] - tb) * self.time_precision, tokens=piece))
                last = cut
            if ends_on_single_ts:
                seek += segment_size                       # nothing spoken after the last timestamp
            else:
                seek += (tokens[last - 1] - tb) * self.input_stride     # resume at the last closed timestamp
        else:
            dur = segment_duration
            stamps = [t for t in tokens if t >= tb]
            if stamps and stamps[-1] != tb:
                dur = (stamps[-1] - tb) * self.time_precision
            out.append(dict(seek=seek, start=time_offset, end=time_offset + dur, tokens=tokens))
            seek += segment_size
        return out, seek, ends_on_single_ts

    # ---- seek loop (:1049-1337)
    def generate_segments(self, features: Union[np.ndarray, DeviceFeatures], tokenizer: Tokenizer,
                          options: TranscriptionOptions, log_progress: bool = False,
                          encoder_output: Optional[EncoderOutput] = None) -> List[Segment]:
        fe = self.feature_extractor
        if not isinstance(features, DeviceFeatures):
            f = np.ascontiguousarray(features, dtype=np.float32)
            slot = self._slot()
            slot.set_features(f)
            features = DeviceFeatures(slot, f.shape[-1])
        content_frames = features.shape[-1] - 1
        if isinstance(options.clip_timestamps, str):
            options.clip_timestamps = [float(t) for t in (options.clip_timestamps.split(",") if options.clip_timestamps else [])]
        points = [round(t * self.frames_per_second) for t in options.clip_timestamps]
        if not points:
            points.append(0)
        if len(points) % 2 == 1:
            points.append(content_frames)
        clips = list(zip(points[::2], points[1::2]))

        idx = 0
        clip_idx = 0
        seek = clips[0][0]
        all_tokens: List[int] = []
        prompt_reset_since = 0
        if options.initial_prompt is not None:
            if isinstance(options.initial_prompt, str):
                all_tokens.extend(tokenizer.encode(" " + options.initial_prompt.strip()))
            else:
                all_tokens.extend(options.initial_prompt)
        all_segments: List[Segment] = []
        last_speech_timestamp = 0.0
        content_duration = float(content_frames * fe.time_per_frame)
        while clip_idx < len(clips):
            clip_start, clip_end = clips[clip_idx]
            clip_end = min(clip_end, content_frames)
            if seek < clip_start:
                seek = clip_start
            if seek >= clip_end:
                clip_idx += 1
                if clip_idx < len(clips):
                    seek = clips[clip_idx][0]
                continue
            time_offset = seek * fe.time_per_frame
            window_end_time = float((seek + fe.nb_max_frames) * fe.time_per_frame)
            segment_size = min(fe.nb_max_frames, content_frames - seek, clip_end - seek)
            segment_duration = segment_size * fe.time_per_frame
            previous_tokens = all_tokens[prompt_reset_since:]
            if seek > 0 or encoder_output is None:
                # pad_or_trim(features[:, seek:seek+segment_size]) happens inside the encoder's window-prep kernel
                encoder_output = self.encode(features, seek=seek, segment_size=segment_size)
            if options.multilingual:
                lang_token, _p = self.model.detect_language(encoder_output)[0][0]
                tokenizer.language = tokenizer.tokenizer.token_to_id(lang_token)
                tokenizer.language_code = lang_token[2:-2]
            prompt = self.get_prompt(tokenizer, previous_tokens, without_timestamps=options.without_timestamps,
                                     prefix=options.prefix if seek == 0 else None, hotwords=options.hotwords)
            result, avg_logprob, temperature, compression_ratio = self.generate_with_fallback(
                encoder_output, prompt, tokenizer, options)
            if options.no_speech_threshold is not None:
                skip = result.no_speech_prob > options.no_speech_threshold
                if options.log_prob_threshold is not None and avg_logprob > options.log_prob_threshold:
                    skip = False                           # confident text beats the no-speech probability
                if skip:
                    seek += segment_size
                    continue
            tokens = result.sequences_ids[0]
            previous_seek = seek
            current, seek, single_ts_ending = self._split_segments_by_timestamps(
                tokenizer=tokenizer, tokens=tokens, time_offset=time_offset, segment_size=segment_size,
                segment_duration=segment_duration, seek=seek)
            if options.word_timestamps:
                # (:1226-1291) cross-attention alignment of this window's text, then the seek / hallucination rules
                # that only exist with word timings
                enc_now = encoder_output

                def align_fn(text_tokens, num_frames, _window):
                    r = self.model.align(enc_now, tokenizer.sot_sequence, [text_tokens], num_frames)[0]
                    pairs = np.asarray(r.alignments, dtype=np.int64).reshape(-1, 2)
                    return pairs[:, 0], pairs[:, 1], np.asarray(r.text_token_probs, dtype=np.float64)   # fp32 values, means in double like np.mean over CT2's Python floats

                # the reference discards add_word_timestamps' return value here (:1227-1235): the last-speech time only
                # moves at the end of this block, so the hallucination rules below still see the PREVIOUS window's
                _wt.add_word_timestamps([current], tokenizer, align_fn, segment_size, self.tokens_per_second,
                                        self.frames_per_second, options.prepend_punctuations,
                                        options.append_punctuations, last_speech_timestamp)
                if not single_ts_ending:
                    lwe = _wt.last_word_end(current)
                    if lwe is not None and lwe > time_offset:
                        seek = round(lwe * self.frames_per_second)
                if options.hallucination_silence_threshold is not None:
                    thr = options.hallucination_silence_threshold
                    first = _wt.next_words_segment(current)
                    if first is not None and _wt.is_segment_anomaly(first):
                        gap = first["start"] - time_offset
                        if gap > thr:                      # leading silence before a probable hallucination: skip it
                            seek = previous_seek + round(gap * self.frames_per_second)
                            continue
                    hal_last_end = last_speech_timestamp
                    for si in range(len(current)):
                        sgm = current[si]
                        if not sgm["words"]:
                            continue
                        if _wt.is_segment_anomaly(sgm):
                            nxt = _wt.next_words_segment(current[si + 1:])
                            hal_next_start = nxt["words"][0]["start"] if nxt is not None else time_offset + segment_duration
                            silence_before = (sgm["start"] - hal_last_end > thr or sgm["start"] < thr
                                              or sgm["start"] - time_offset < 2.0)
                            silence_after = (hal_next_start - sgm["end"] > thr or _wt.is_segment_anomaly(nxt)
                                             or window_end_time - sgm["end"] < 2.0)
                            if silence_before and silence_after:
                                seek = round(max(time_offset + 1, sgm["start"]) * self.frames_per_second)
                                if content_duration - sgm["end"] < thr:
                                    seek = content_frames
                                current[si:] = []
                                break
                        hal_last_end = sgm["end"]
                lwe = _wt.last_word_end(current)
                if lwe is not None:
                    last_speech_timestamp = lwe
            for sg in current:
                text = tokenizer.decode(sg["tokens"])
                if sg["start"] == sg["end"] or not text.strip():
                    continue
                all_tokens.extend(sg["tokens"])
                idx += 1
                all_segments.append(Segment(id=idx, seek=previous_seek, start=sg["start"], end=sg["end"], text=text,
                                            tokens=sg["tokens"], temperature=temperature, avg_logprob=avg_logprob,
                                            compression_ratio=compression_ratio, no_speech_prob=result.no_speech_prob,
                                            words=([Word(**w) for w in sg["words"]] if options.word_timestamps else None)))
            if not options.condition_on_previous_text or temperature > options.prompt_reset_on_temperature:
                prompt_reset_since = len(all_tokens)
        return all_segments

    # ---- temperature fallback (:1350-1478)
    def generate_with_fallback(self, encoder_output: EncoderOutput, prompt: List[int], tokenizer: Tokenizer,
                               options: TranscriptionOptions):
        max_initial_timestamp_index = int(round(options.max_initial_timestamp / self.time_precision))
        max_length = len(prompt) + options.max_new_tokens if options.max_new_tokens is not None else self.max_length
        if max_length > self.max_length:
            raise ValueError(
                f"The length of the prompt is {len(prompt)}, and the `max_new_tokens` {max_length - len(prompt)}. Thus, the "
                f"combined length of the prompt and `max_new_tokens` is: {max_length}. This exceeds the `max_length` of the "
                f"Whisper model: {self.max_length}. You should either reduce the length of your prompt, or reduce the value "
                f"of `max_new_tokens`, so that their combined length is less that {self.max_length}.")
        tried = []
        under_cr = []
        chosen = None
        temperature = 0.0
        for temperature in options.temperatures:
            if temperature > 0:
                kw = dict(beam_size=1, num_hypotheses=options.best_of, sampling_topk=0, sampling_temperature=temperature)
            else:
                kw = dict(beam_size=options.beam_size, patience=options.patience)
            result = self.model.generate(
                encoder_output, [prompt], length_penalty=options.length_penalty,
                repetition_penalty=options.repetition_penalty, no_repeat_ngram_size=options.no_repeat_ngram_size,
                max_length=max_length, return_scores=True, return_no_speech_prob=True,
                suppress_blank=options.suppress_blank, suppress_tokens=options.suppress_tokens,
                max_initial_timestamp_index=max_initial_timestamp_index, **kw)[0]
            tokens = result.sequences_ids[0]
            n = len(tokens)
            cum_logprob = result.scores[0] * (n ** options.length_penalty)       # CT2 score -> sum of log-probs
            avg_logprob = cum_logprob / (n + 1)
            text = tokenizer.decode(tokens).strip()
            cr = get_compression_ratio(text)
            chosen = (result, avg_logprob, temperature, cr)
            tried.append(chosen)
            retry = False
            if options.compression_ratio_threshold is not None:
                if cr > options.compression_ratio_threshold:
                    retry = True                            # too repetitive
                else:
                    under_cr.append(chosen)
            if options.log_prob_threshold is not None and avg_logprob < options.log_prob_threshold:
                retry = True                                # too unlikely
            if (options.no_speech_threshold is not None and result.no_speech_prob > options.no_speech_threshold
                    and options.log_prob_threshold is not None and avg_logprob < options.log_prob_threshold):
                retry = False                               # silence: accept as is
            if not retry:
                break
        else:
            best = max(under_cr or tried, key=lambda r: r[1])
            chosen = (best[0], best[1], temperature, best[3])   # report the LAST temperature (prompt-reset rule)
        return chosen

    # ---- prompt (:1480-1513)
    def get_prompt(self, tokenizer: Tokenizer, previous_tokens: List[int], without_timestamps: bool = False,
                   prefix: Optional[str] = None, hotwords: Optional[str] = None) -> List[int]:
        half = self.max_length // 2
        prompt: List[int] = []
        if previous_tokens or (hotwords and not prefix):
            prompt.append(tokenizer.sot_prev)
            if hotwords and not prefix:
                hw = tokenizer.encode(" " + hotwords.strip())
                prompt.extend(hw[: half - 1] if len(hw) >= half else hw)
            if previous_tokens:
                prompt.extend(previous_tokens[-(half - 1):])
        prompt.extend(tokenizer.sot_sequence)
        if without_timestamps:
            prompt.append(tokenizer.no_timestamps)
        if prefix:
            px = tokenizer.encode(" " + prefix.strip())
            if len(px) >= half:
                px = px[: half - 1]
            if not without_timestamps:
                prompt.append(tokenizer.timestamp_begin)
            prompt.extend(px)
        return prompt

    # ---- language detection (:1716-1789)
    def detect_language(self, audio: Optional[np.ndarray] = None, features: Optional[Union[np.ndarray, DeviceFeatures]] = None,
                        vad_filter: bool = False, vad_parameters: Union[dict, VadOptions, None] = None,
                        language_detection_segments: int = 1, language_detection_threshold: float = 0.5, _seek: int = 0
                        ) -> Tuple[str, float, List[Tuple[str, float]]]:
        assert audio is not None or features is not None, "Either `audio` or `features` must be provided."
        fe = self.feature_extractor
        if audio is not None:
            if vad_filter:
                if isinstance(vad_parameters, dict):
                    vad_parameters = VadOptions(**vad_parameters)
                chunks, _ = _vad.collect_chunks(audio, _vad.get_speech_timestamps(audio, vad_parameters, model=self._vad_model()))
                audio = np.concatenate(chunks, axis=0)
            audio = audio[: language_detection_segments * fe.n_samples]
            slot = self._slot()
            features = DeviceFeatures(slot, slot.logmel(np.ascontiguousarray(audio, dtype=np.float32)))
        elif not isinstance(features, DeviceFeatures):
            f = np.ascontiguousarray(features, dtype=np.float32)
            slot = self._slot()
            slot.set_features(f)
            features = DeviceFeatures(slot, f.shape[-1])
        total = min(features.n_frames - _seek, language_detection_segments * fe.nb_max_frames)
        votes: Dict[str, List[float]] = {}
        all_language_probs: List[Tuple[str, float]] = []
        language, language_probability = "en", 0.0
        decided = False
        for i in range(0, max(total, 1), fe.nb_max_frames):
            enc = self.encode(features, seek=_seek + i, segment_size=min(fe.nb_max_frames, total - i))
            results = self.model.detect_language(enc)[0]
            all_language_probs = [(tok[2:-2], p) for tok, p in results]
            language, language_probability = all_language_probs[0]
            if language_probability > language_detection_threshold:
                decided = True
                break
            votes.setdefault(language, []).append(language_probability)
        if not decided and votes:
            language = max(votes, key=lambda k: len(votes[k]))      # majority vote over the windows
            language_probability = max(votes[language])
        return language, language_probability, all_language_probs
