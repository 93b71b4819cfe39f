/*
 * wlx.h — C-ABI of libwlx.so, the MI355X-native (gfx950) Whisper streaming-inference engine.
 *
 * This is the drop-in boundary #3 of SURVEY.md §8(b): it replaces the five call sites through
 * which collabora/WhisperLive reaches its un-vendored numerical engines (faster-whisper /
 * CTranslate2), each entry point citing the reference interface it stands in for
 * (paths relative to the reference checkout):
 *
 *   wlx_logmel            <- faster_whisper FeatureExtractor.__call__
 *                            (whisper_live/transcriber/transcriber_faster_whisper.py:655,862;
 *                             whisper_live/batch_inference.py:258; recipe restated in-tree at
 *                             whisper_live/transcriber/tensorrt_utils.py:177-190)
 *   wlx_encode            <- ctranslate2.models.Whisper.encode
 *                            (transcriber_faster_whisper.py:1339-1348; batch_inference.py:270-271)
 *   wlx_generate          <- ctranslate2.models.Whisper.generate
 *                            (transcriber_faster_whisper.py:1394-1407; batch_inference.py:355-357)
 *   wlx_detect_language   <- ctranslate2.models.Whisper.detect_language
 *                            (transcriber_faster_whisper.py:1140,1771; batch_inference.py:283)
 *   wlx_features_set/get  <- ctranslate2.StorageView.from_array
 *                            (transcriber_faster_whisper.py:1820-1823)
 *
 * Conventions: plain pointers and sizes only (no torch / C++ types); every function returns
 * 0 on success and a non-zero wlx_status otherwise (no exceptions cross the boundary, no
 * callbacks into the host language); wlx_last_error() returns a thread-local message.
 * Threading: concurrent calls are safe on DISTINCT slots (each slot owns a HIP stream and all
 * of its scratch); calls on the same slot are to be serialised by the caller, which is what the
 * reference does (one transcription thread per client, faster_whisper_backend.py:121; or the
 * single batch-worker thread, batch_inference.py:155-187). The library enforces it: a second call
 * on a busy slot returns WLX_ERR_STATE instead of running, wlx_slot_destroy waits for the call in
 * flight, slot ids are never reused. No entry point uses the legacy (null) HIP stream, so slot
 * creation and destruction are safe while other slots are decoding.
 * NULL-stream caveat for embedding applications: by default a slot's stream is created with
 * hipExtStreamCreateWithCUMask (all CUs enabled) so that it owns a hardware queue (the first
 * WLX_DEDICATED_QUEUES = 4 live slots of a device). That constructor takes no flags:
 * the stream is a BLOCKING stream (hipStreamGetFlags == 0), i.e. it synchronises implicitly with
 * the legacy NULL stream of the process. Work that the embedding process issues on the NULL stream
 * of the same device (a framework's default stream, a synchronous hipMemcpy) therefore serialises
 * with every slot, and if it is issued while a slot captures its decode-step graph that one
 * wlx_generate call fails with WLX_ERR_HIP (the slot's stream is replaced, the next call works).
 * Such processes should set WLX_SLOT_CU_MASK=off (ordinary non-blocking streams on the shared queue
 * pool) — and get exactly that by DEFAULT (round 5) once any weight tensor has been handed to
 * wlx_engine_create with on_device = 1, i.e. when the process demonstrably holds device memory of
 * another runtime; an explicit WLX_SLOT_CU_MASK always wins. The flag is process-wide and sticky; a
 * slot created BEFORE the first such engine gives its blocking stream back at its next call
 * (round 6), so no creation order has to be observed. The library states the mode on stderr at the
 * first slot creation in each mode (WLX_QUIET silences it). wlx_slot_create estimates the slot's
 * device memory first and refuses (WLX_ERR_NOMEM, with the figure) what the device cannot hold.
 */
#ifndef WLX_H
#define WLX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WLX_ABI_VERSION 1

typedef enum {
    WLX_OK = 0,
    WLX_ERR_ARG = 1,      /* bad argument / shape */
    WLX_ERR_HIP = 2,      /* a HIP runtime call failed */
    WLX_ERR_WEIGHT = 3,   /* missing / mis-shaped weight tensor */
    WLX_ERR_STATE = 4,    /* call order (e.g. generate before encode) */
    WLX_ERR_NOMEM = 5
} wlx_status;

/* Whisper architecture. n_audio_ctx = 1500, n_text_ctx = 448, head_dim = 64. */
typedef struct {
    int32_t n_mels;       /* 80 | 128 */
    int32_t d_model;
    int32_t n_heads;
    int32_t enc_layers;
    int32_t dec_layers;
    int32_t ffn;
    int32_t vocab;
    int32_t n_audio_ctx;  /* 1500 */
    int32_t n_text_ctx;   /* 448  */
} wlx_spec;

/* One weight tensor, named with the Hugging Face Whisper state-dict key (e.g.
 * "model.encoder.layers.0.self_attn.q_proj.weight"), **float32 only**, C-contiguous, in host or device
 * memory (there is no dtype field: a half-precision checkpoint must be widened by the caller — the Python binding,
 * whisperlive_amd/engine.py, does `tensor.to(float32)` for fp16 / bf16 torch tensors, a transient device copy of that ONE
 * tensor list; the values are unchanged, the engine rounds projection matrices to fp16 itself). The engine copies/repacks
 * every tensor into its own MFMA-fragment layout during create; the caller may free the tensors afterwards. */
typedef struct {
    const char* name;
    const void* data;
    int32_t ndim;
    int64_t shape[4];
    int32_t on_device;    /* 0 = host pointer, 1 = HIP device pointer on `device` */
} wlx_tensor;

typedef struct wlx_engine wlx_engine;

/* Token ids the decoding rules need; resolved by NAME on the host side from tokenizer.json
 * (never hard-coded). */
typedef struct {
    int32_t sot, eot, no_timestamps, timestamp_begin, no_speech, blank /* id of " " or -1 */;
} wlx_token_ids;

/* Decoding options = the keyword arguments of ctranslate2 Whisper.generate as the reference
 * passes them (transcriber_faster_whisper.py:1380-1407). */
typedef struct {
    int32_t beam_size;                 /* T=0: 5 ; T>0: 1 */
    float   patience;                  /* 1 */
    int32_t num_hypotheses;            /* T=0: 1 ; T>0: best_of=5 */
    float   length_penalty;            /* 1 */
    float   repetition_penalty;        /* 1 */
    int32_t no_repeat_ngram_size;      /* 0 */
    int32_t max_length;                /* prompt + generated, <= 448 */
    int32_t suppress_blank;            /* bool */
    const int32_t* suppress_tokens;    /* may be NULL */
    int32_t n_suppress_tokens;
    int32_t max_initial_timestamp_index; /* 50 */
    int32_t sampling_topk;             /* 0 = whole vocabulary (only value used by the reference for T>0); 1 = greedy */
    float   sampling_temperature;      /* 0 -> beam search */
    uint64_t seed;                     /* counter-based RNG seed for T>0 */
    wlx_token_ids ids;
} wlx_gen_opts;

/* Per-stage GPU times of the last calls on a slot, milliseconds (HIP events on the slot stream). */
typedef struct {
    float logmel_ms, encode_ms, generate_ms;
    int32_t decode_steps;
} wlx_timings;

int32_t wlx_abi_version(void);
const char* wlx_last_error(void);

int32_t wlx_engine_create(const wlx_spec* spec, const wlx_tensor* weights, int32_t n_weights,
                          int32_t device, wlx_engine** out);
void    wlx_engine_destroy(wlx_engine* e);
int32_t wlx_engine_spec(const wlx_engine* e, wlx_spec* out);

/* A slot = one unit of concurrency: its own HIP stream plus every device buffer a call needs
 * (feature ring, encoder activations, cross-attention K/V, self-attention KV cache, beam state),
 * sized for `max_batch` audio items (<= 64) and `max_rows_per_item` decoder rows (beams, <= 16) per item;
 * max_batch * max_rows_per_item <= 320 decoder rows per step (64 clips x beam 5: the reference's worker takes any
 * max_batch_size, whisper_live/batch_inference.py:113-121 — wider batches than a slot holds are decoded as consecutive
 * groups over the same resident encoder output by the host side). */
int32_t wlx_slot_create(wlx_engine* e, int32_t max_batch, int32_t max_rows_per_item, int32_t* slot_out);
int32_t wlx_slot_destroy(wlx_engine* e, int32_t slot);

/* log-mel of `n` float32 PCM samples (host pointer, 16 kHz mono) for item `item` of the slot.
 * Result stays on the device as float32 [n_mels, n_frames], n_frames = (n+160)/160. */
int32_t wlx_logmel(wlx_engine* e, int32_t slot, int32_t item, const float* pcm, int64_t n,
                   int32_t* n_frames_out);
/* The same in two halves, for callers that keep the stream's PCM resident in HBM (the device-side
 * counterpart of the session buffer of whisper_live/backend/base.py:173-234): wlx_pcm_put copies
 * `n` host samples into the item's device PCM buffer; wlx_logmel_resident computes the features of
 * whatever is resident. wlx_logmel == wlx_pcm_put + wlx_logmel_resident.
 * Neither waits for the device, and the feature launches of a slot's items are RECORDED and issued together — one launch of each
 * kernel for all requested items — in front of the first call that consumes them (wlx_encode, wlx_features_get, wlx_features_set,
 * wlx_timings_get, wlx_sync) or that replaces a requested item's PCM: n_frames_out is a function of n alone, errors of the launch
 * itself surface at that later call. */
int32_t wlx_pcm_put(wlx_engine* e, int32_t slot, int32_t item, const float* pcm, int64_t n);
int32_t wlx_logmel_resident(wlx_engine* e, int32_t slot, int32_t item, int32_t* n_frames_out);
/* Copy an item's device features to host / replace them from host (float32 [n_mels, n_frames]). */
int32_t wlx_features_get(wlx_engine* e, int32_t slot, int32_t item, float* out, int64_t cap_floats,
                         int32_t* n_frames_out);
int32_t wlx_features_set(wlx_engine* e, int32_t slot, int32_t item, const float* feats,
                         int32_t n_mels, int32_t n_frames);

/* Encoder forward for items 0..batch-1. Item i uses feature frames [seek[i], seek[i]+seg[i]),
 * zero-padded (in log-mel space) to 3000 = pad_or_trim (transcriber_faster_whisper.py:1125-1127).
 * Leaves encoder output [batch,1500,d] and the cross-attention K/V of every decoder layer on the device. */
int32_t wlx_encode(wlx_engine* e, int32_t slot, int32_t batch, const int32_t* seek, const int32_t* seg);
/* float32 copy of the encoder output of one item, [1500, d_model]. */
int32_t wlx_encoder_output_get(wlx_engine* e, int32_t slot, int32_t item, float* out, int64_t cap_floats);

/* Autoregressive decode for items 0..batch-1 (item i uses prompt i). Outputs, per item and
 * hypothesis h < num_hypotheses (best first): generated token ids (prompt and EOT excluded),
 * their count, the CT2-style score (sum of log-probs incl. EOT / len^length_penalty), and per
 * item the no-speech probability (softmax prob. of ids.no_speech at the sot position).
 * The call returns when the results are final, which can be BEFORE the slot's stream is idle: the search kernels store the
 * results in pinned host memory and order them before the "done" word the host polls, so a decode that ends on an end-of-text
 * does not wait for the one step that was already enqueued behind the finish (~0.4 ms on Whisper-small). Later calls on the
 * slot are ordered behind it on the stream; wlx_timings_get / wlx_sync wait for it. */
int32_t wlx_generate(wlx_engine* e, int32_t slot, int32_t batch,
                     const int32_t* prompts, const int32_t* prompt_lens, int32_t prompt_stride,
                     const wlx_gen_opts* opts,
                     int32_t* tokens_out, int32_t tokens_stride /* per hypothesis */,
                     int32_t* n_tokens_out, float* scores_out, float* no_speech_prob_out);

/* The same with an item map: decoder item i attends to the encoder output of item enc_items[i]
 * (NULL = identity). Lets the batched fallback of whisper_live/batch_inference.py:318-384 retry a subset
 * of a batch without re-encoding it (the reference re-encodes, :334-339). */
int32_t wlx_generate_ex(wlx_engine* e, int32_t slot, int32_t batch, const int32_t* enc_items,
                        const int32_t* prompts, const int32_t* prompt_lens, int32_t prompt_stride,
                        const wlx_gen_opts* opts,
                        int32_t* tokens_out, int32_t tokens_stride,
                        int32_t* n_tokens_out, float* scores_out, float* no_speech_prob_out);

/* One decoder step on [sot]; softmax restricted to `lang_ids`; probs_out[batch][n_lang]. */
int32_t wlx_detect_language(wlx_engine* e, int32_t slot, int32_t batch, int32_t sot,
                            const int32_t* lang_ids, int32_t n_lang, float* probs_out);

/* Device times of the slot's last log-mel / encode / generate (HIP events); waits for whichever of them is still running. */
int32_t wlx_timings_get(wlx_engine* e, int32_t slot, wlx_timings* out);
int32_t wlx_sync(wlx_engine* e, int32_t slot);

/* ---- word timestamps (PRODUCT entry point: transcribe(word_timestamps=True) calls it once per segment group) ---- */
/* Word alignment — replaces ctranslate2.models.Whisper.align(encoder_output, start_sequence, text_tokens, num_frames,
 * median_filter_width) (whisper_live/transcriber/transcriber_faster_whisper.py:1657-1663).
 * tokens = start_sequence (n_sot ids) + [no_timestamps] + text_tokens + [eot]  (n_tokens <= 448) of encoder item `item`;
 * heads = n_heads (layer, head) pairs (the model's alignment heads). Outputs: the DTW path as parallel arrays
 * text_indices / time_indices (n_path <= path_cap entries; time in encoder positions of 20 ms) and
 * text_token_probs[n_tokens - n_sot - 2] = softmax over ids < eot of each text token. */
int32_t wlx_align(wlx_engine* e, int32_t slot, int32_t item, const int32_t* tokens, int32_t n_tokens, int32_t n_sot,
                  int32_t num_frames, int32_t median_filter_width, const int32_t* heads, int32_t n_heads, int32_t eot,
                  int32_t* text_indices, int32_t* time_indices, int32_t path_cap, int32_t* n_path_out,
                  float* text_token_probs);

/* ---- voice-activity probabilities (Silero VAD, 16 kHz) — PRODUCT entry points: the VAD gate of every use_vad session ----
 * Replaces the model call inside faster_whisper.vad.get_speech_timestamps (onnxruntime, one CPU thread) that the
 * reference makes before every transcription when the client asks for VAD:
 * whisper_live/transcriber/transcriber_faster_whisper.py:830-838 and, per batch item, whisper_live/batch_inference.py:245-248;
 * I/O contract of the same network: whisper_live/vad.py:50-109 (512-sample windows, 64 samples of left context, zero
 * initial state, one probability per window). The hysteresis / padding / chunk bookkeeping that turns probabilities
 * into sample ranges stays on the host (whisperlive_amd/vad.py).
 * Weights: HOST pointers, float32 row-major, copied and repacked at creation:
 *   stft_basis [258,256]; enc_w[i] [Cout,Cin,3], enc_b[i] [Cout] with (Cin,Cout) = (129,128) (128,64) (64,64) (64,128);
 *   lstm_w_ih, lstm_w_hh [512,128] and lstm_b_ih, lstm_b_hh [512] in gate order i,f,g,o; out_w [128]; out_b [1]. */
typedef struct wlx_vad wlx_vad;
typedef struct {
    const float* stft_basis;
    const float* enc_w[4];
    const float* enc_b[4];
    const float* lstm_w_ih;
    const float* lstm_w_hh;
    const float* lstm_b_ih;
    const float* lstm_b_hh;
    const float* out_w;
    const float* out_b;
} wlx_vad_weights;
int32_t wlx_vad_create(const wlx_vad_weights* w, int32_t device, wlx_vad** out);
void    wlx_vad_destroy(wlx_vad* v);
/* `n` float32 samples (host pointer) -> ceil(n/512) probabilities (host buffer of `cap` floats); the last window is
 * zero-padded. Each call starts from a zero state, like get_speech_timestamps on a fresh chunk. Thread-safe (calls on
 * one object are serialised; it owns its HIP stream). device_ms_out (nullable): kernel time between HIP events. */
int32_t wlx_vad_probs(wlx_vad* v, const float* pcm, int64_t n, float* probs_out, int32_t cap,
                      int32_t* n_windows_out, float* device_ms_out);

/* ---- device-resident PCM ring of ONE client stream (round 6; PRODUCT entry points of the streaming path) ----
 * The device-side mirror of the session buffer ServeClientBase keeps on the host (whisper_live/backend/base.py:173-234:
 * `frames_np`, the 45 s cap / 30 s trim of add_frames :191-198, the chunk taken by get_audio_chunk_for_processing :219-234).
 * Every packet a client sends crosses PCIe ONCE — wlx_ring_append — and everything that reads audio afterwards reads HBM:
 * the VAD gate (wlx_vad_probs_resident) and the log-mel front end, whose PCM -> LDS loads walk the list of speech ranges the
 * gate kept (wlx_logmel_ring: replaces faster_whisper.vad.collect_chunks + np.concatenate + a second upload,
 * transcriber_faster_whisper.py:836-838,862). A ring is independent of slots: the socket thread appends while the
 * transcription thread reads; calls on one ring are serialised by the library.
 * Sample positions are ABSOLUTE stream positions (sample 0 = the first sample ever appended): a trim moves `base`,
 * never the positions a caller already holds. A range that has been trimmed away fails with WLX_ERR_STATE. */
typedef struct wlx_ring wlx_ring;
int32_t wlx_ring_create(wlx_engine* e, int64_t capacity_samples /* 0: 64 s */, wlx_ring** out);
void    wlx_ring_destroy(wlx_ring* r);
/* add_frames (base.py:173-234): if more than `max_resident` samples are resident, the OLDEST `trim` samples are dropped
 * first (45 s / 30 s in the reference: pass 720000 / 480000; max_resident <= 0 disables the rule), then the `n` host
 * samples are appended. Outputs (nullable): samples dropped by this call, first resident position, resident count.
 * The copy is complete on return. */
int32_t wlx_ring_append(wlx_ring* r, const float* samples, int64_t n, int64_t max_resident, int64_t trim,
                        int64_t* dropped_out, int64_t* base_out, int64_t* resident_out);
int32_t wlx_ring_state(wlx_ring* r, int64_t* base_out, int64_t* resident_out);
/* wlx_vad_probs on ring samples [start, start + n): no host-to-device copy. Same contract otherwise (zero initial
 * state, ceil(n / 512) windows, the last one zero-padded) plus `extra_zero_windows` all-zero windows behind them
 * (faster_whisper.vad pads n to the NEXT multiple of 512 — a whole window of zeros when n already is one — and the
 * host side asks for the same count, so the two paths segment identically). `v` and `r` must live on the same device. */
int32_t wlx_vad_probs_resident(wlx_vad* v, wlx_ring* r, int64_t start, int64_t n, int32_t extra_zero_windows,
                               float* probs_out, int32_t cap, int32_t* n_windows_out, float* device_ms_out);
/* Host only (no device work): the hysteresis segmentation of per-window speech probabilities into padded sample ranges —
 * faster_whisper.vad.get_speech_timestamps' loop (reference call site: transcriber_faster_whisper.py:825-852), statement for
 * statement whisperlive_amd/vad.py speech_segments_from_probs. It runs between the VAD launch and the log-mel launch, with the
 * GPU waiting. thr / neg: the thresholds rounded to float32; the durations in samples as the Python code computes them
 * (max_speech may be +inf). start_end_out: [cap][2]. */
int32_t wlx_vad_segments(const float* probs, int32_t n_windows, int64_t n_samples, double thr, double neg, double min_speech,
                         double pad, double max_speech, double min_silence, double min_silence_at_max,
                         int64_t* start_end_out, int32_t cap, int32_t* n_out);
/* log-mel of the CONCATENATION of `n_ranges` ring ranges [ranges[2 i], ranges[2 i + 1]) (absolute positions, ascending,
 * <= 256 of them) into item `item` of the slot: identical features to wlx_logmel on the concatenated samples. The launch
 * is issued at once (it reads the ring, which the socket thread may trim later). */
int32_t wlx_logmel_ring(wlx_engine* e, int32_t slot, int32_t item, wlx_ring* r, const int64_t* ranges, int32_t n_ranges,
                        int32_t* n_frames_out);

/* ==== everything below: TEST / PROFILING hooks (used only by tests/, scripts/ and bench.py's roofline leg; not part of
 * the drop-in boundary; the product entry points end here) ================================================================= */
/* next-token logits [rows, vocab] of the last decoder step executed on the slot */
int32_t wlx_debug_logits_get(wlx_engine* e, int32_t slot, float* out, int32_t rows, int64_t cap_floats);
/* teacher-forced decoder pass: feed `n` tokens of one sequence (item 0), return logits [n, vocab] */
int32_t wlx_debug_decode_logits(wlx_engine* e, int32_t slot, const int32_t* tokens, int32_t n, float* out);
/* run the search kernels on caller-supplied logits (float32 [steps][rows][vocab], host):
 * exercises logits processors + beam/sampling bookkeeping without the network */
int32_t wlx_debug_search(wlx_engine* e, int32_t slot, const float* logits, int32_t steps,
                         const int32_t* prompt, int32_t prompt_len, const wlx_gen_opts* opts,
                         int32_t* tokens_out, int32_t tokens_stride, int32_t* n_tokens_out, float* scores_out);
/* time `iters` replays of one full decode step (rows x vocab GEMV chain) with HIP events; returns avg ms */
int32_t wlx_debug_time_decode_step(wlx_engine* e, int32_t slot, int32_t rows, int32_t t, int32_t iters,
                                   float* avg_ms_out);

/* per-kernel HIP-event profile of one decode step (eager launches bracketed by event pairs on the slot
 * stream), aggregated by kernel name; bytes_per_launch = algorithmic bytes (weights / K,V streamed once). */
typedef struct {
    char name[64];
    float launches_per_step, avg_us, total_us_per_step;
    double bytes_per_launch;
} wlx_kernel_stat;
/* In-kernel timeline of one decode step. Only libwlx_trace.so (the same sources built with -DWLX_TRACE) records;
 * the production library returns WLX_ERR_STATE. out: [n_launches][(2048 + 1) * 8] u64 (n_launches <= 320), names: [n_launches][48]. */
int32_t wlx_debug_trace_step(wlx_engine* e, int32_t slot, int32_t rows, int32_t t, int32_t with_search,
                             uint64_t* out, int64_t cap_u64, char* names, int32_t* n_launches_out);

int32_t wlx_debug_profile_step(wlx_engine* e, int32_t slot, int32_t rows, int32_t t, int32_t iters,
                               wlx_kernel_stat* out, int32_t cap, int32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* WLX_H */
