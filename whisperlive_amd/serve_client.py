"""Per-client streaming session on the HIP engine — the sibling of ``ServeClientFasterWhisper``
(whisper_live/backend/faster_whisper_backend.py:14) under the ``ServeClientBase`` contract
(whisper_live/backend/base.py:11): same constructor arguments, same JSON messages, same buffer / offset / commit
rules, so an unchanged ``TranscriptionServer`` (whisper_live/server.py:288-314) can select it as a backend.

``ServeClientBase`` here restates the session state machine of base.py (buffer cap 45 s / trim 30 s :173-203, chunk
slicing :216-234, clip rule :205-214, the transcription loop and its latency timer :88-137, segment commit logic
:383-483) — it is the caller that defines the chunk shapes and the metric, not a kernel target.
``ServeClientHIP.transcribe_audio`` is row 5 of SURVEY.md §8a.

Differences on purpose (hazards listed in SURVEY.md §5, not behaviour): the shared-model lock is taken with a
context manager (the reference leaks it on exception, faster_whisper_backend.py:234-246); with one engine per GPU and
a slot per client thread the lock is not needed for correctness at all and is only honoured when
``serialize_single_model=True`` is requested.
"""
from __future__ import annotations

import json
import os
import logging
import queue
import threading
import time
from typing import List, Optional

import numpy as np

from . import metrics as wl_metrics
from . import vad as _vad


class ServeClientBase:
    RATE = 16000
    SERVER_READY = "SERVER_READY"
    DISCONNECT = "DISCONNECT"
    MAX_BUFFER_DURATION_S = 45
    BUFFER_TRIM_DURATION_S = 30
    CLIP_THRESHOLD_DURATION_S = 25
    CLIP_TAIL_DURATION_S = 5
    FIRST_FRAME_WAIT_TIMEOUT_S = 0.1
    MAX_TRANSCRIPT_LENGTH = 500
    MAX_TRANSLATION_QUEUE_SIZE = 100

    def __init__(self, client_uid, websocket, send_last_n_segments=10, no_speech_thresh=0.45, clip_audio=False,
                 same_output_threshold=10, translation_queue=None, diarization=None, word_timestamps=False):
        self.client_uid = client_uid
        self.websocket = websocket
        self.send_last_n_segments = send_last_n_segments
        self.no_speech_thresh = no_speech_thresh
        self.clip_audio = clip_audio
        self.same_output_threshold = same_output_threshold
        self.translation_queue = translation_queue
        self.diarization = diarization
        self.word_timestamps = word_timestamps
        self.segment_post_processor = None
        # session state
        self.frames_np: Optional[np.ndarray] = None
        self.frames_offset = 0.0          # seconds of audio dropped from the front of the buffer
        self.timestamp_offset = 0.0       # seconds of the session already committed
        self.text: List[str] = []
        self.transcript: List[dict] = []
        self.current_out = ""
        self.prev_out = ""
        self.same_output_count = 0
        self.end_time_for_same_output = None
        self.exit = False
        self.language = None
        self.lock = threading.Lock()
        self.frames_ready = threading.Event()
        self._wake = threading.Event()    # set by cleanup(): the loop's pauses (the reference's time.sleep calls) end at once

    # ---- audio buffer ------------------------------------------------------------------------------------------
    def add_frames(self, frame_np: np.ndarray):
        with self.lock:
            buf = self.frames_np
            trimmed = 0
            if buf is not None and buf.shape[0] > self.MAX_BUFFER_DURATION_S * self.RATE:
                self.frames_offset += float(self.BUFFER_TRIM_DURATION_S)
                trimmed = int(self.BUFFER_TRIM_DURATION_S * self.RATE)
                buf = buf[trimmed:]
                if self.timestamp_offset < self.frames_offset:      # nothing was committed in the dropped audio
                    self.timestamp_offset = self.frames_offset
            self.frames_np = frame_np.copy() if buf is None else np.concatenate((buf, frame_np), axis=0)
            self._frames_appended(frame_np, trimmed)
        self.frames_ready.set()

    def _frames_appended(self, frame_np: np.ndarray, trimmed: int):
        """Hook, called under `self.lock` right after `frames_np` took the packet (`trimmed` samples were dropped from its front
        first): backends that keep a device-side mirror of the buffer update it here."""

    def clip_audio_if_no_valid_segment(self):
        with self.lock:
            start = int((self.timestamp_offset - self.frames_offset) * self.RATE)
            if self.frames_np[start:].shape[0] > self.CLIP_THRESHOLD_DURATION_S * self.RATE:
                total = self.frames_np.shape[0] / self.RATE
                self.timestamp_offset = self.frames_offset + total - self.CLIP_TAIL_DURATION_S

    def get_audio_chunk_for_processing(self):
        with self.lock:
            take = max(0, (self.timestamp_offset - self.frames_offset) * self.RATE)
            chunk = self.frames_np[int(take):].copy()
            # where the chunk starts in the stream (absolute sample position): what a device-side mirror of the buffer is indexed by
            self._chunk_abs = (int(round(self.frames_offset * self.RATE)) + int(take), chunk.shape[0])
        return chunk, chunk.shape[0] / self.RATE

    def get_audio_chunk_duration(self, input_bytes) -> float:
        return input_bytes.shape[0] / self.RATE

    # ---- transcription loop (defines the metric sample) ------------------------------------------------------
    def speech_to_text(self):
        while not self.exit:
            if self.frames_np is None:
                while self.frames_np is None and not self.exit:
                    self.frames_ready.wait(timeout=self.FIRST_FRAME_WAIT_TIMEOUT_S)
                continue
            if self.clip_audio:
                self.clip_audio_if_no_valid_segment()
            chunk, duration = self.get_audio_chunk_for_processing()
            if duration < 1.0:
                self._pause(0.1)
                continue
            try:
                sample = chunk.copy()
                t0 = time.time()
                result = self.transcribe_audio(sample)
                if result is None or self.language is None:
                    self.timestamp_offset += duration          # no voice activity in this chunk
                    self._pause(0.25)
                    continue
                wl_metrics.track_transcription_latency(time.time() - t0)
                wl_metrics.track_audio_processed(duration)
                self.handle_transcription_output(result, duration)
            except Exception as e:  # noqa: BLE001 — the reference logs and carries on (base.py:134-137)
                logging.error(f"[ERROR]: Failed to transcribe audio chunk: {e}")
                wl_metrics.track_error("transcription")
                time.sleep(0.01)
        self.on_transcription_thread_exit()
        logging.info("Exiting speech to text thread")

    def _resident(self, input_sample):
        """Hook: the chunk in the form the backend's transcriber prefers (ServeClientHIP: a ResidentAudio over its device PCM ring)."""
        return input_sample

    def _transcribe_locked(self, input_sample, kw):
        input_sample = self._resident(input_sample)
        if self.serialize:
            with ServeClientHIP.SINGLE_MODEL_LOCK:
                return self.transcriber.transcribe(input_sample, **kw)
        return self.transcriber.transcribe(input_sample, **kw)                 # concurrent: own slot / HIP stream

    def _downgrade_vad(self, e):
        logging.warning(f"use_vad requested by {self.client_uid} but unavailable: {e}")
        self.use_vad = False
        try:
            self.websocket.send(json.dumps({"uid": self.client_uid, "status": "WARNING",
                                            "message": "use_vad ignored: no Silero VAD weights are configured on this server"}))
        except Exception:  # noqa: BLE001 — a closed socket must not hide the transcript path
            pass

    def on_transcription_thread_exit(self):
        """Hook run by the transcription thread as it ends (backends release per-thread resources here)."""

    def transcribe_audio(self, input_sample):
        raise NotImplementedError

    def handle_transcription_output(self, result, duration):
        raise NotImplementedError

    # ---- formatting / sending ----------------------------------------------------------------------------------
    def format_segment(self, start, end, text, completed=False, speaker=None, words=None):
        seg = {"start": "{:.3f}".format(start), "end": "{:.3f}".format(end), "text": text, "completed": completed}
        if speaker is not None:
            seg["speaker"] = speaker
        if words is not None:
            seg["words"] = words
        return seg

    def prepare_segments(self, last_segment=None):
        segs = self.transcript[-self.send_last_n_segments:].copy() if len(self.transcript) >= self.send_last_n_segments \
            else self.transcript.copy()
        if last_segment is not None:
            segs = segs + [last_segment]
        return segs

    def send_transcription_to_client(self, segments):
        if self.segment_post_processor is not None:
            done = []
            for seg in segments:
                try:
                    r = self.segment_post_processor(seg)
                    done.append(seg if r is None else r)
                except Exception as e:  # noqa: BLE001
                    logging.error(f"[ERROR]: segment_post_processor failed: {e}")
                    done.append(seg)
            segments = done
        try:
            self.websocket.send(json.dumps({"uid": self.client_uid, "segments": segments}))
            for seg in segments:
                wl_metrics.track_segment_emitted(completed=seg.get("completed", False))
        except Exception as e:  # noqa: BLE001
            logging.error(f"[ERROR]: Sending data to client: {e}")

    def disconnect(self):
        self.websocket.send(json.dumps({"uid": self.client_uid, "message": self.DISCONNECT}))

    def cleanup(self):
        logging.info("Cleaning up.")
        self.exit = True
        self.frames_ready.set()
        self._wake.set()

    def _pause(self, seconds: float):
        """The reference sleeps here (base.py:117,128,445); same duration, but a session being torn down does not linger
        in it holding its engine slot."""
        self._wake.wait(seconds)

    # ---- segment accessors (backends name these fields differently) ------------------------------------------
    def get_segment_no_speech_prob(self, segment):
        return getattr(segment, "no_speech_prob", 0)

    def get_segment_start(self, segment):
        return getattr(segment, "start", getattr(segment, "start_ts", 0))

    def get_segment_end(self, segment):
        return getattr(segment, "end", getattr(segment, "end_ts", 0))

    def _identify_speaker(self, segment):
        if self.diarization is None or self.frames_np is None:
            return None
        try:
            s0 = int(self.get_segment_start(segment) * self.RATE)
            s1 = int(self.get_segment_end(segment) * self.RATE)
            base = max(0, int((self.timestamp_offset - self.frames_offset) * self.RATE))
            piece = self.frames_np[base + s0: base + s1]
            if len(piece) < self.RATE * 0.3:
                return None
            return self.diarization.identify_speaker(piece, self.RATE)
        except Exception as e:  # noqa: BLE001
            logging.error(f"Diarization error: {e}")
            return None

    def _extract_words(self, segment, time_offset):
        if not self.word_timestamps:
            return None
        words = getattr(segment, "words", None)
        if not words:
            return None
        return [{"word": w.word, "start": "{:.3f}".format(time_offset + w.start), "end": "{:.3f}".format(time_offset + w.end),
                 "probability": round(w.probability, 4)} for w in words]

    def _queue_translation(self, seg: dict):
        if self.translation_queue:
            try:
                self.translation_queue.put(seg.copy(), timeout=0.1)
            except queue.Full:
                logging.warning("Translation queue is full, skipping segment")

    # ---- commit logic ------------------------------------------------------------------------------------------
    def update_segments(self, segments, duration):
        """All segments but the last are committed (if their no-speech probability allows); the last one is sent as
        in-progress until the SAME text has come back more than `same_output_threshold` times, at which point it is
        committed up to the time it was first repeated. Committing advances `timestamp_offset`."""
        advance = None
        self.current_out = ""
        last_segment = None
        tail = segments[-1]
        tail_ok = self.get_segment_no_speech_prob(tail) <= self.no_speech_thresh

        if len(segments) > 1 and tail_ok:
            for s in segments[:-1]:
                self.text.append(s.text)
                with self.lock:
                    start = self.timestamp_offset + self.get_segment_start(s)
                    end = self.timestamp_offset + min(duration, self.get_segment_end(s))
                if start >= end or self.get_segment_no_speech_prob(s) > self.no_speech_thresh:
                    continue
                done = self.format_segment(start, end, s.text, completed=True, speaker=self._identify_speaker(s),
                                           words=self._extract_words(s, self.timestamp_offset))
                self.transcript.append(done)
                self._queue_translation(done)
                advance = min(duration, self.get_segment_end(s))

        if tail_ok:
            self.current_out += tail.text
            words = self._extract_words(tail, self.timestamp_offset)
            with self.lock:
                last_segment = self.format_segment(self.timestamp_offset + self.get_segment_start(tail),
                                                   self.timestamp_offset + min(duration, self.get_segment_end(tail)),
                                                   self.current_out, completed=False, words=words)

        if self.current_out != "" and self.current_out.strip() == self.prev_out.strip():
            self.same_output_count += 1
            if self.end_time_for_same_output is None:          # remember when the repetition started
                self.end_time_for_same_output = self.get_segment_end(tail)
            self._pause(0.1)                                   # base.py:445 — wait briefly for new voice activity
        else:
            self.same_output_count = 0
            self.end_time_for_same_output = None

        if self.same_output_count > self.same_output_threshold:
            if not self.text or self.text[-1].strip().lower() != self.current_out.strip().lower():
                self.text.append(self.current_out)
                with self.lock:
                    done = self.format_segment(self.timestamp_offset,
                                               self.timestamp_offset + min(duration, self.end_time_for_same_output),
                                               self.current_out, completed=True)
                    self.transcript.append(done)
                    self._queue_translation(done)
            self.current_out = ""
            advance = min(duration, self.end_time_for_same_output)
            self.same_output_count = 0
            last_segment = None
            self.end_time_for_same_output = None
        else:
            self.prev_out = self.current_out

        if advance is not None:
            with self.lock:
                self.timestamp_offset += advance
        self._trim_transcript()
        return last_segment

    def _trim_transcript(self):
        if len(self.transcript) > self.MAX_TRANSCRIPT_LENGTH:
            self.transcript = self.transcript[-self.MAX_TRANSCRIPT_LENGTH:]
        if len(self.text) > self.MAX_TRANSCRIPT_LENGTH:
            self.text = self.text[-self.MAX_TRANSCRIPT_LENGTH:]


class ServeClientHIP(ServeClientBase):
    """Backend adaptor for the MI355X engine. Class-level registries mirror the reference's SINGLE_MODEL /
    BATCH_WORKER hooks (faster_whisper_backend.py:15-17) but are keyed by GPU: one shared transcriber per device."""

    MODELS = {}                     # device index -> WhisperModelHIP (one engine = one copy of the weights per GPU)
    MODELS_LOCK = threading.Lock()
    SINGLE_MODEL_LOCK = threading.Lock()
    BATCH_WORKER = None             # optional whisperlive_amd.batching.BatchInferenceWorker (all devices)
    BATCH_WORKERS = {}              # device index -> BatchInferenceWorker (one per GPU; set by TranscriptionServer)
    BACKEND_NAME = "faster_whisper"  # the stock Python client only keeps completed segments for this string (client.py:182,399)

    def __init__(self, websocket, task="transcribe", device=None, language=None, client_uid=None, model="small.en",
                 initial_prompt=None, vad_parameters=None, use_vad=True, single_model=True, send_last_n_segments=10,
                 no_speech_thresh=0.45, clip_audio=False, same_output_threshold=7, cache_path="~/.cache/whisper-live/",
                 translation_queue=None, hotwords=None, diarization=None, word_timestamps=False, *,
                 device_index: int = 0, transcriber=None, model_factory=None, serialize_single_model: bool = False,
                 start_thread: bool = True, max_batch: int = 1):
        super().__init__(client_uid, websocket, send_last_n_segments, no_speech_thresh, clip_audio, same_output_threshold,
                         translation_queue, diarization, word_timestamps)
        self.cache_path = cache_path
        self.model_size_or_path = model
        self.language = "en" if (model or "").endswith("en") else language
        self.task = task
        self.initial_prompt = initial_prompt
        self.vad_parameters = vad_parameters or {"threshold": 0.5}
        self.hotwords = hotwords
        self.compute_type = "float16"
        self.device_index = device_index
        self._ring = None
        self.serialize = serialize_single_model
        if model is None and transcriber is None:
            return
        try:
            if transcriber is not None:
                self.transcriber = transcriber
            else:
                with ServeClientHIP.MODELS_LOCK:
                    key = (device_index, model) if not single_model else device_index
                    if key not in ServeClientHIP.MODELS:
                        if model_factory is not None:
                            ServeClientHIP.MODELS[key] = model_factory(model, device_index)
                        else:
                            # a transcriber that a batch worker will drive needs slots as wide as the worker's batches
                            ServeClientHIP.MODELS[key] = self.create_model(model, device_index, max_batch=max_batch)
                    self.transcriber = ServeClientHIP.MODELS[key]
        except Exception as e:  # noqa: BLE001 — same client-visible behaviour as faster_whisper_backend.py:108-116
            logging.error(f"Failed to load model: {e}")
            self.websocket.send(json.dumps({"uid": self.client_uid, "status": "ERROR",
                                            "message": f"Failed to load model: {str(self.model_size_or_path)}"}))
            self.websocket.close()
            return
        self.use_vad = use_vad
        self.trans_thread = threading.Thread(target=self.speech_to_text, daemon=True)
        if start_thread:
            self.trans_thread.start()
        self.websocket.send(json.dumps({"uid": self.client_uid, "message": self.SERVER_READY,
                                        "backend": self.BACKEND_NAME}))

    @staticmethod
    def create_model(model: str, device_index: int, max_batch: int = 1):
        from .transcriber import WhisperModelHIP
        return WhisperModelHIP(model, device="cuda", device_index=device_index, compute_type="float16",
                               max_batch=max(1, min(int(max_batch), 64)))

    def set_language(self, info):
        if info.language_probability > 0.5:
            self.language = info.language
            logging.info(f"Detected language {self.language} with probability {info.language_probability}")
            self.websocket.send(json.dumps({"uid": self.client_uid, "language": self.language,
                                            "language_prob": info.language_probability}))

    # ---- device-resident PCM ring (round 6): the packets cross PCIe once, in add_frames; VAD and log-mel read HBM ----------------
    _ring = None                           # None: not created yet; False: off for this session; else the PcmRing (whisperlive_amd/engine.py)

    def _frames_appended(self, frame_np, trimmed):
        """Mirror the session buffer on the transcriber's GPU (whisperlive_amd.engine.PcmRing: the same 45 s cap / 30 s trim, decided
        HERE and checked there). Any failure switches the mirror off for the session: the host path needs nothing from it."""
        if self._ring is False:
            return
        try:
            if self._ring is None:
                eng = getattr(getattr(self, "transcriber", None), "engine", None)
                make = getattr(eng, "create_ring", None)
                if make is None or os.environ.get("WLX_PCM_RING", "1") == "0" or (ServeClientHIP.BATCH_WORKER or ServeClientHIP.BATCH_WORKERS):
                    self._ring = False
                    return
                self._ring = make()
                self._ring_expect = 0
            dropped, base, resident = self._ring.append(frame_np)
            self._ring_expect += frame_np.shape[0] - dropped
            if dropped != trimmed or resident != self.frames_np.shape[0] or base != int(round(self.frames_offset * self.RATE)):
                raise RuntimeError(f"ring out of step with frames_np: dropped {dropped} vs {trimmed}, resident {resident} vs "
                                   f"{self.frames_np.shape[0]}, base {base} vs {self.frames_offset * self.RATE}")
        except Exception as e:  # noqa: BLE001
            logging.warning(f"device PCM ring disabled for {self.client_uid}: {e}")
            ring, self._ring = self._ring, False
            if ring:
                try:
                    ring.close()
                except Exception:  # noqa: BLE001
                    pass

    def _resident(self, input_sample):
        """`input_sample` as a ResidentAudio when it is the chunk get_audio_chunk_for_processing just took and the ring holds it."""
        ring, ca = self._ring, getattr(self, "_chunk_abs", None)
        if not ring or ca is None or ca[1] != input_sample.shape[0]:
            return input_sample
        from .transcriber import ResidentAudio
        return ResidentAudio(ring, ca[0], ca[1], input_sample)

    def transcribe_audio(self, input_sample):
        worker = ServeClientHIP.BATCH_WORKER or ServeClientHIP.BATCH_WORKERS.get(self.device_index)
        if worker is not None:
            from .batching import BatchRequest
            req = BatchRequest(audio=input_sample, language=self.language, task=self.task,
                               initial_prompt=self.initial_prompt, use_vad=self.use_vad,
                               vad_parameters=self.vad_parameters if self.use_vad else None,
                               word_timestamps=self.word_timestamps, client_uid=self.client_uid)
            worker.submit(req)
            req.future.wait(timeout=30)
            if isinstance(req.error, _vad.VadUnavailable) and self.use_vad:
                # the worker's transcriber has no Silero weights (a model_factory-built one): same downgrade as the direct path
                # below — tell the client ONCE, run ungated from now on, resubmit this chunk (re-raising on every
                # chunk left the session alive but silent)
                self._downgrade_vad(req.error)
                req = BatchRequest(audio=input_sample, language=self.language, task=self.task, initial_prompt=self.initial_prompt,
                                   use_vad=False, vad_parameters=None, word_timestamps=self.word_timestamps, client_uid=self.client_uid)
                worker.submit(req)
                req.future.wait(timeout=30)
            if req.error:
                raise req.error
            if self.language is None and req.info is not None:
                self.set_language(req.info)
            return req.result
        kw = dict(initial_prompt=self.initial_prompt, language=self.language, task=self.task, vad_filter=self.use_vad,
                  vad_parameters=self.vad_parameters if self.use_vad else None, hotwords=self.hotwords,
                  word_timestamps=self.word_timestamps)
        try:
            result, info = self._transcribe_locked(input_sample, kw)
        except _vad.VadUnavailable as e:
            # use_vad reached a transcriber with no Silero weights (a model_factory-built one: the server's own availability
            # check only covers transcribers it builds itself). Same outcome as there: tell the client ONCE, run ungated —
            # raising here on every chunk would leave the session alive but silent.
            self._downgrade_vad(e)
            kw.update(vad_filter=False, vad_parameters=None)
            result, info = self._transcribe_locked(input_sample, kw)
        if self.language is None and info is not None:
            self.set_language(info)
        return result

    def on_transcription_thread_exit(self):
        release = getattr(type(self.transcriber), "release_slot", None) if hasattr(self, "transcriber") else None
        if release is not None:
            self.transcriber.release_slot()            # the engine slot goes back to the per-GPU pool
        with self.lock:
            ring, self._ring = self._ring, False
        if ring:
            ring.close()

    def handle_transcription_output(self, result, duration):
        segments = []
        if len(result):
            self.t_start = None
            segments = self.prepare_segments(self.update_segments(result, duration))
        if len(segments):
            self.send_transcription_to_client(segments)
