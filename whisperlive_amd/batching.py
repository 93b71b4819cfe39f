"""Cross-client micro-batcher on the HIP engine — the role of ``BatchInferenceWorker``
(whisper_live/batch_inference.py:87-450; ``BatchRequest`` :51-84): session threads submit requests and block on an
event; ONE worker thread collects up to ``max_batch_size`` requests within ``batch_window_ms`` and runs them as one
batched encode + one batched generate per temperature, with per-item temperature fallback.

Same queueing / fallback / result semantics as the reference, including its quirks that change results (only the
first 30 s of each item is used :259; the trailing pad frame is kept :258-259; fallback temperatures run greedy
because CTranslate2's default ``sampling_topk`` is 1 :343-357; ``suppress_tokens`` come from the first item's
tokenizer :316). MI355X-first differences that do not change results: log-mel runs on the device per item
(the reference's serial CPU stage :236-263), language detection is evaluated once per batch instead of once per
unknown-language item (:283), and a retry re-uses the encoder output already resident in HBM through an item map
(the reference re-encodes the failed items :334-339).
"""
from __future__ import annotations

import logging
import queue
import threading
import time
from dataclasses import dataclass, field
from math import ceil
from typing import Any, Dict, List, Optional

import numpy as np

from . import vad as _vad
from .tokenizer import Tokenizer
from .transcriber import get_compression_ratio, get_suppressed_tokens, pad_or_trim
from .types import Segment, TranscriptionInfo


@dataclass
class BatchRequest:
    audio: np.ndarray
    language: Optional[str] = None
    task: str = "transcribe"
    initial_prompt: Optional[str] = None
    use_vad: bool = True
    vad_parameters: Optional[Dict] = None
    word_timestamps: bool = False
    client_uid: Optional[str] = None
    future: threading.Event = field(default_factory=threading.Event)
    result: Optional[Any] = None
    info: Optional[Any] = None
    error: Optional[Exception] = None


class BatchInferenceWorker:
    TEMPERATURES = (0.0, 0.2, 0.4, 0.6, 0.8, 1.0)
    COMPRESSION_RATIO_THRESHOLD = 2.4
    LOGPROB_THRESHOLD = -1.0
    NO_SPEECH_THRESHOLD = 0.6

    def __init__(self, transcriber, max_batch_size: int = 8, batch_window_ms: int = 50, lanes: int = 2,
                 wait_when_idle: bool = False):
        self.transcriber = transcriber
        # never collect more requests than one engine slot can hold (WhisperModelHIP.max_batch); a duck-typed or mocked
        # transcriber without that integer keeps the configured size
        cap = getattr(transcriber, "max_batch", None)
        if isinstance(cap, int) and not isinstance(cap, bool) and 1 <= cap < max_batch_size:
            logging.warning(f"[BatchInference] max_batch_size {max_batch_size} clamped to the transcriber's slot width {cap}")
            max_batch_size = cap
        self.max_batch_size = max_batch_size
        self.batch_window_ms = batch_window_ms
        # MI355X-first: `lanes` worker threads instead of the reference's one (whisper_live/batch_inference.py:118-121). Batches
        # are COLLECTED by one lane at a time, exactly as the reference collects them (first request, then up to
        # max_batch_size within batch_window_ms), but a second lane may collect and run the next batch while the first is
        # still on the GPU: every lane decodes on its own engine slot / hardware queue, so a request that arrives just after
        # a batch started no longer waits for that whole batch (configs[2] through the worker: p50 45 -> see DESIGN.md §5).
        # lanes=1 is the reference's behaviour.
        self.lanes = max(1, int(lanes))
        # The collection window exists to let more requests join a batch — which only pays while the GPU is busy with an
        # earlier batch (rows that join a decode are almost free, a batch that waits behind another is not). With every lane
        # idle the first request of a batch therefore starts at once unless more requests are ALREADY queued; the reference
        # always waits out its window (:140-153; wait_when_idle=True restores that).
        self.wait_when_idle = bool(wait_when_idle)
        self._busy = 0
        self._busy_lock = threading.Lock()
        self._queue: "queue.Queue[BatchRequest]" = queue.Queue()
        self._stop_event = threading.Event()
        self._collect_lock = threading.Lock()
        self._thread: Optional[threading.Thread] = None
        self._threads: List[threading.Thread] = []

    def start(self):
        self._threads = [threading.Thread(target=self._worker_loop, daemon=True, name=f"wlx-batch-lane{i}") for i in range(self.lanes)]
        self._thread = self._threads[0]
        for t in self._threads:
            t.start()
        logging.info(f"[BatchInference] Started (max_batch={self.max_batch_size}, window={self.batch_window_ms}ms, lanes={self.lanes})")

    def stop(self):
        self._stop_event.set()
        for t in (self._threads or ([self._thread] if self._thread else [])):
            t.join(timeout=5)

    def submit(self, request: BatchRequest):
        self._queue.put(request)

    # ---- collection -------------------------------------------------------------------------------------------
    def _collect(self) -> List[BatchRequest]:
        """one batch, collected as the reference collects it (:126-153); only one lane collects at a time"""
        with self._collect_lock:
            try:
                batch = [self._queue.get(timeout=0.5)]
            except queue.Empty:
                return []
            with self._busy_lock:
                idle = self._busy == 0
            window = self.batch_window_ms / 1000.0
            if idle and not self.wait_when_idle:
                window = 0.0                          # nothing on the GPU: take what is already queued, wait for nothing
            deadline = time.monotonic() + window
            while len(batch) < self.max_batch_size:
                if window == 0.0:
                    try:
                        batch.append(self._queue.get_nowait())
                        continue
                    except queue.Empty:
                        break
                left = deadline - time.monotonic()
                if left <= 0:
                    break
                try:
                    batch.append(self._queue.get(timeout=left))
                except queue.Empty:
                    break
            with self._busy_lock:
                self._busy += 1                       # counted before the next lane may start collecting
            return batch

    def _worker_loop(self):
        try:
            while not self._stop_event.is_set():
                batch = self._collect()
                if not batch:
                    continue
                try:
                    self._process_batch(batch)
                except Exception as e:  # noqa: BLE001 — the worker must survive (tests/test_batch_inference.py:99-120)
                    logging.error(f"[BatchInference] Batch processing error: {e}")
                    for req in batch:
                        if not req.future.is_set():
                            req.error = e
                            req.future.set()
                finally:
                    with self._busy_lock:
                        self._busy -= 1
        finally:
            release = getattr(type(self.transcriber), "release_slot", None)
            if release is not None:
                self.transcriber.release_slot()        # this lane's engine slot goes back to the pool

    def _process_batch(self, batch: List[BatchRequest]):
        if len(batch) == 1:
            self._process_single(batch[0])
        else:
            logging.info(f"[BatchInference] Processing batch of {len(batch)}")
            self._process_multi(batch)

    def _process_single(self, req: BatchRequest):
        try:
            result, info = self.transcriber.transcribe(req.audio, language=req.language, task=req.task,
                                                       initial_prompt=req.initial_prompt, vad_filter=req.use_vad,
                                                       vad_parameters=req.vad_parameters if req.use_vad else None)
            req.result = list(result) if result is not None else []
            req.info = info
        except Exception as e:  # noqa: BLE001
            req.error = e
        finally:
            req.future.set()

    # ---- batched path -----------------------------------------------------------------------------------------
    def _encode_batch(self, audios: List[np.ndarray]):
        """Device path when the transcriber is a WhisperModelHIP (per-item log-mel kernels + one batched encoder
        launch, nothing returns to the host); generic duck-typed path otherwise (what the reference does)."""
        tr = self.transcriber
        if getattr(type(tr), "encode_audio_batch", None) is not None:     # class-level: a MagicMock transcriber has "every" attribute
            return tr.encode_audio_batch(audios)
        feats = np.stack([pad_or_trim(tr.feature_extractor(a)) for a in audios])
        return tr.encode(feats)

    def _process_multi(self, batch: List[BatchRequest]):
        tr = self.transcriber
        sr = tr.feature_extractor.sampling_rate
        ready = []
        for req in batch:
            try:
                audio = req.audio
                if req.use_vad:
                    params = req.vad_parameters or {}
                    opts = _vad.VadOptions(**params) if isinstance(params, dict) else params
                    vm = getattr(type(tr), "_vad_model", None)        # class-level: a MagicMock transcriber has "every" attribute
                    chunks = _vad.get_speech_timestamps(audio, opts, model=tr._vad_model() if vm is not None else None)
                    if chunks:
                        pieces, _ = _vad.collect_chunks(audio, chunks)
                        audio = np.concatenate(pieces, axis=0) if pieces else audio
                if audio.shape[0] == 0:
                    req.result = []
                    req.info = self._make_info(req, 0.0, 0.0)
                    req.future.set()
                    continue
                ready.append((req, audio, audio.shape[0] / sr))
            except Exception as e:  # noqa: BLE001
                req.error = e
                req.future.set()
        if not ready:
            return
        try:
            enc = self._encode_batch([a for _, a, _ in ready])
            n = len(ready)
            lang_results = None
            toks: List[Tokenizer] = []
            prompts: List[List[int]] = []
            langs: List[str] = []
            for i, (req, _audio, _dur) in enumerate(ready):
                lang = req.language
                if lang is None:
                    try:
                        if lang_results is None:
                            lang_results = tr.model.detect_language(enc)
                        if lang_results and len(lang_results) > i and lang_results[i]:
                            lang = lang_results[i][0][0].strip("<|>")
                    except Exception:  # noqa: BLE001 — reference falls back to English (:288-289)
                        lang = "en"
                langs.append(lang or "en")
                tk = Tokenizer(tr.hf_tokenizer, tr.model.is_multilingual, task=req.task, language=lang or "en")
                prev = tk.encode(" " + req.initial_prompt.strip()) if req.initial_prompt else []
                toks.append(tk)
                prompts.append(tr.get_prompt(tk, previous_tokens=prev, without_timestamps=False))
            suppress = get_suppressed_tokens(toks[0], [-1])
            final: List[Optional[tuple]] = [None] * n
            pending = list(range(n))
            for temp in self.TEMPERATURES:
                if not pending:
                    break
                sub_enc = enc if len(pending) == n else self._select(enc, pending, [ready[i][1] for i in pending])
                results = tr.model.generate(
                    sub_enc, [prompts[i] for i in pending], beam_size=5 if temp == 0.0 else 1, patience=1,
                    length_penalty=1, max_length=tr.max_length, suppress_blank=True, suppress_tokens=suppress,
                    return_scores=True, return_no_speech_prob=True, sampling_temperature=temp, repetition_penalty=1,
                    no_repeat_ngram_size=0)
                still = []
                for j, idx in enumerate(pending):
                    g = results[j]
                    tokens = g.sequences_ids[0]
                    ln = len(tokens)
                    avg_logprob = (g.scores[0] * ln) / (ln + 1) if ln > 0 else 0.0
                    text = toks[idx].decode(tokens).strip()
                    cr = get_compression_ratio(text) if text else 0.0
                    bad = cr > self.COMPRESSION_RATIO_THRESHOLD or avg_logprob < self.LOGPROB_THRESHOLD
                    silent = g.no_speech_prob > self.NO_SPEECH_THRESHOLD and avg_logprob < self.LOGPROB_THRESHOLD
                    if not bad or silent or temp == self.TEMPERATURES[-1]:
                        final[idx] = (g, avg_logprob, temp)
                    else:
                        still.append(idx)
                pending = still
            for i, (req, _audio, duration) in enumerate(ready):
                try:
                    g, avg_logprob, used_temp = final[i]
                    subs, _, _ = tr._split_segments_by_timestamps(
                        tokenizer=toks[i], tokens=g.sequences_ids[0], time_offset=0,
                        segment_size=int(ceil(duration) * tr.frames_per_second), segment_duration=duration, seek=0)
                    segs = []
                    for k, sub in enumerate(subs):
                        text = toks[i].decode(sub["tokens"]).strip()
                        if not text:
                            continue
                        segs.append(Segment(id=k, seek=sub.get("seek", 0), start=sub["start"], end=sub["end"], text=text,
                                            tokens=sub["tokens"], avg_logprob=avg_logprob,
                                            compression_ratio=get_compression_ratio(text),
                                            no_speech_prob=g.no_speech_prob, words=None, temperature=used_temp))
                    req.result = segs
                    req.info = self._make_info(req, duration, duration, language=langs[i])
                except Exception as e:  # noqa: BLE001
                    req.error = e
                finally:
                    req.future.set()
        except Exception as e:  # noqa: BLE001
            logging.error(f"[BatchInference] GPU batch error: {e}")
            for req, *_ in ready:
                if not req.future.is_set():
                    req.error = e
                    req.future.set()

    def _select(self, enc, indices: List[int], audios: List[np.ndarray]):
        """Encoder output restricted to `indices`: an item map over what is already in HBM when the transcriber
        supports it, otherwise a re-encode of those items (the reference's behaviour)."""
        if hasattr(enc, "select"):
            return enc.select(indices)
        return self._encode_batch(audios)

    def _make_info(self, req, duration, duration_after_vad, language=None):
        return TranscriptionInfo(language=language or req.language or "en", language_probability=1.0, duration=duration,
                                 duration_after_vad=duration_after_vad, all_language_probs=None,
                                 transcription_options=None, vad_options=None)
