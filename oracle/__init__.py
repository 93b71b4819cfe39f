"""oracle/ — CPU restatement of the reference's per-chunk Whisper hot path. TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the timed CPU baseline — never as the product path. The product
(``whisperlive_amd``) fails loudly when the HIP library is missing; it has no CPU fallback.

What is restated, and from where (paths relative to the reference checkout /root/reference):

* ``logmel.py``   — faster_whisper.feature_extractor.FeatureExtractor.__call__ (faster-whisper==1.2.0,
  requirements/server.txt:1; un-vendored) as called at
  whisper_live/transcriber/transcriber_faster_whisper.py:862; recipe restated in-tree at
  whisper_live/transcriber/tensorrt_utils.py:177-190; pad_or_trim at tensorrt_utils.py:80-104.
* ``model.py``    — the Whisper network that ctranslate2.models.Whisper.encode / .generate evaluate
  (CTranslate2 v4.x, docker/Dockerfile.rocm:20; un-vendored); definition followed:
  transformers models/whisper/modeling_whisper.py (conv stem :566-567,618-619; sinusoids :55;
  pre-LN layers :360-402, :416-498; q scaling :267,309; tied projection :965-970).
* ``decoding.py`` — the search inside ctranslate2.models.Whisper.generate as the reference drives it
  (transcriber_faster_whisper.py:1380-1407): logits processors restated from the published OpenAI
  definition (whisper/decoding.py: SuppressBlank, SuppressTokens, ApplyTimestampRules; identical to
  transformers generation/logits_process.py:1909-2047), beam search / sampling per the CT2 contract
  the caller relies on (score = sum logp incl. EOT / len^length_penalty, :1409-1414).
* ``alignment.py`` — ctranslate2.models.Whisper.align as called at transcriber_faster_whisper.py:1657-1663 (word
  timestamps): published openai/whisper timing.py algorithm; median filter and DTW pinned against transformers
  generation_whisper.py (tests/golden/align_golden.npz).
* ``silero_vad.py`` — the Silero-VAD probability network faster_whisper.vad.get_speech_timestamps evaluates through
  onnxruntime before every VAD-gated transcription (transcriber_faster_whisper.py:830-838; batch_inference.py:245-248);
  I/O contract whisper_live/vad.py:50-109; layer stack from the published model description.
  PARITY UNPINNED (no weights / runtime offline): pinned only against an independent torch build of the same stack.
  The segmentation bookkeeping around it (hysteresis, padding, time map) is host logic in whisperlive_amd/vad.py.

PARITY PINNING STATUS. The reference's own tests pin NO tensor on this path (every test
that touches the transcriber mocks it; the only result-level pin is one WER<5% sentence that needs
model weights and a FLAC decoder, neither available offline), and faster-whisper / CTranslate2 /
onnxruntime cannot be imported here. The oracle is therefore pinned against the independent
implementation that IS importable in this container — Hugging Face ``transformers`` Whisper
(encoder states, next-token logits, WhisperTimeStampLogitsProcessor, mel filterbank,
WhisperFeatureExtractor) on seeded random weights; the generating script is
``tests/golden/make_golden.py`` and its outputs are committed under ``tests/golden/``.
Beam-search tie-breaking and the T>0 RNG stream of CTranslate2 itself remain **parity unpinned**.
"""
