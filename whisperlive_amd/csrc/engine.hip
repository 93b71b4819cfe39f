// engine.hip — C-ABI (include/wlx.h) over the gfx950 kernels: weight ingestion/repacking, slots
// (one HIP stream + all scratch per concurrent stream), and the host orchestration of
// log-mel -> encoder -> prefill -> hipGraph-replayed decode steps.
#include "engine.h"
#include <sched.h>
#include <atomic>
#include <limits>
#include <algorithm>
#include <cmath>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>

using namespace wlx;

// one polite spin-wait iteration (the decode loop polls a pinned word): the ISA's spin hint where there is one
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int wlx::set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define CK(call)                                                                                       \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(WLX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CKR(call)                                 \
    do {                                          \
        int r_ = (call);                          \
        if (r_ != WLX_OK) return r_;              \
    } while (0)

extern "C" int32_t wlx_abi_version(void) { return WLX_ABI_VERSION; }
extern "C" const char* wlx_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// Nothing in this library may touch the legacy (null) stream once slots exist: while ANY stream is capturing a decode
// graph, a legacy-stream operation from another thread (hipMemset, synchronous hipMemcpy, hipDeviceSynchronize) fails with
// "would make the legacy stream depend on a capturing ... stream" AND invalidates that capture — i.e. a second client
// connecting (slot creation) used to be able to break the first client's transcription. Set-up work that is not tied to a
// slot therefore runs on a per-device non-blocking utility stream and waits for it explicitly.
hipStream_t wlx::util_stream() {
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    auto it = streams.find(dev);
    if (it != streams.end()) return it->second;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    streams[dev] = st;
    return st;
}
static int upload_sync(void* dst, const void* src, size_t bytes) {      // host -> device, complete on return
    hipStream_t us = wlx::util_stream();
    if (!us) return fail(WLX_ERR_HIP, "utility stream creation failed");
    CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, us));
    CK(hipStreamSynchronize(us));
    return WLX_OK;
}

// ------------------------------------------------------------------------------------------------
// allocation helpers
template <typename T>
static int dalloc(std::vector<void*>& pool, T** out, size_t count, bool zero = true) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(WLX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    if (zero) {
        hipStream_t us = wlx::util_stream();
        if (!us) return fail(WLX_ERR_HIP, "utility stream creation failed");
        e = hipMemsetAsync(p, 0, bytes, us);
        if (e == hipSuccess) e = hipStreamSynchronize(us);
        if (e != hipSuccess) return fail(WLX_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    pool.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return WLX_OK;
}

// ------------------------------------------------------------------------------------------------
// weight ingestion
struct WeightSource {
    std::map<std::string, const wlx_tensor*> by_name;
    float* staging = nullptr;   // device fp32 staging for host tensors
    size_t staging_cap = 0;
    hipStream_t stream = nullptr;

    const wlx_tensor* find(const std::string& n) const {
        auto it = by_name.find(n);
        return it == by_name.end() ? nullptr : it->second;
    }
    static size_t numel(const wlx_tensor* t) {
        size_t n = 1;
        for (int i = 0; i < t->ndim; ++i) n *= (size_t)t->shape[i];
        return n;
    }
    // device fp32 view of a tensor (valid until the next call when the tensor lives on the host)
    int device_f32(const wlx_tensor* t, const float** out) {
        if (t->on_device) { *out = reinterpret_cast<const float*>(t->data); return WLX_OK; }
        size_t n = numel(t);
        if (n > staging_cap) {
            if (staging) (void)hipFree(staging);
            staging = nullptr;
            CK(hipMalloc(reinterpret_cast<void**>(&staging), n * sizeof(float)));
            staging_cap = n;
        }
        CK(hipStreamSynchronize(stream));  // previous consumer of the staging buffer
        CKR(upload_sync(staging, t->data, n * sizeof(float)));
        *out = staging;
        return WLX_OK;
    }
};

static int need(WeightSource& ws, const std::string& name, std::initializer_list<int64_t> shape,
                const wlx_tensor** out) {
    const wlx_tensor* t = ws.find(name);
    if (!t) return fail(WLX_ERR_WEIGHT, "missing weight '%s'", name.c_str());
    if (t->ndim != (int)shape.size()) return fail(WLX_ERR_WEIGHT, "weight '%s': ndim %d", name.c_str(), t->ndim);
    int i = 0;
    for (int64_t s : shape) {
        if (t->shape[i] != s)
            return fail(WLX_ERR_WEIGHT, "weight '%s': dim %d is %lld, expected %lld", name.c_str(), i,
                        (long long)t->shape[i], (long long)s);
        ++i;
    }
    *out = t;
    return WLX_OK;
}

// copy an fp32 vector into engine memory at dst+offset (dst pre-allocated, zeroed)
static int load_vec(WeightSource& ws, const std::string& name, int64_t n, float* dst) {
    const wlx_tensor* t;
    CKR(need(ws, name, {n}, &t));
    if (t->on_device) CK(hipMemcpyAsync(dst, t->data, n * sizeof(float), hipMemcpyDeviceToDevice, ws.stream));
    else CKR(upload_sync(dst, t->data, n * sizeof(float)));
    return WLX_OK;
}
static int alloc_vec(Engine* e, WeightSource& ws, const std::string& name, int64_t n, float** out) {
    CKR(dalloc(e->allocs, out, (size_t)n));
    return load_vec(ws, name, n, *out);
}
// pack W[N][K] into a packed image (already allocated, [NT_total][KT]) at n-tile offset nt0
static int pack_into(WeightSource& ws, const std::string& name, int64_t N, int64_t K, half_t* Wp, int KT, int nt0) {
    const wlx_tensor* t;
    CKR(need(ws, name, {N, K}, &t));
    const float* src;
    CKR(ws.device_f32(t, &src));
    launch_pack_linear(src, (int)N, (int)K, K, Wp, KT, nt0, ws.stream);
    CK(hipGetLastError());
    return WLX_OK;
}
static int alloc_packed(Engine* e, int64_t N, int64_t K, half_t** out, int* KT_out) {
    int KT = (int)((K + 31) / 32);
    int NT = (int)((N + 15) / 16);
    CKR(dalloc(e->allocs, out, (size_t)NT * KT * 512));
    *KT_out = KT;
    return WLX_OK;
}

// Slaney mel filterbank exactly as faster-whisper's FeatureExtractor.get_mel_filters builds it
// (librosa.filters.mel(sr=16000, n_fft=400, n_mels, htk=False, norm="slaney"); float64 math,
// float32 storage). Equals HF audio_utils.mel_filter_bank(201, n, 0, 8000, 16000, "slaney", "slaney").
static void build_mel_filters(int n_mels, std::vector<float>& w, std::vector<int>& range) {
    const int nb = 201;
    std::vector<double> fft(nb), melf(n_mels + 2);
    for (int i = 0; i < nb; ++i) fft[i] = (double)i * 8000.0 / 200.0;
    const double max_mel = 45.245640471924965, f_sp = 200.0 / 3.0, min_log_hz = 1000.0;
    const double min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    const double step = max_mel / (double)(n_mels + 1);   // np.linspace(0, max_mel, n_mels + 2)
    for (int i = 0; i < n_mels + 2; ++i) {
        double m = (i == n_mels + 1) ? max_mel : (double)i * step;
        melf[i] = (m >= min_log_mel) ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
    }
    w.assign((size_t)n_mels * nb, 0.f);
    range.assign((size_t)n_mels * 2, 0);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = melf[i + 1] - melf[i], fd1 = melf[i + 2] - melf[i + 1];
        const double enorm = 2.0 / (melf[i + 2] - melf[i]);
        int lo = nb, hi = 0;
        for (int b = 0; b < nb; ++b) {
            const double lower = -(melf[i] - fft[b]) / fd0, upper = (melf[i + 2] - fft[b]) / fd1;
            const double v = std::max(0.0, std::min(lower, upper));
            const float f = (float)((double)(float)v * enorm);
            w[(size_t)i * nb + b] = f;
            if (f != 0.f) { lo = std::min(lo, b); hi = std::max(hi, b + 1); }
        }
        if (lo >= hi) { lo = 0; hi = 0; }
        range[2 * i] = lo; range[2 * i + 1] = hi;
    }
}

static int build_logmel_consts(Engine* e) {
    std::vector<float> win(400), tw(800), filt;
    std::vector<int> range;
    const double PI2 = 6.283185307179586476925286766559;
    for (int i = 0; i < 400; ++i) {
        win[i] = (float)(0.5 - 0.5 * std::cos(PI2 * (double)i / 400.0));   // np.hanning(401)[:-1]
        tw[2 * i] = (float)std::cos(PI2 * (double)i / 400.0);
        tw[2 * i + 1] = (float)std::sin(PI2 * (double)i / 400.0);
    }
    build_mel_filters(e->spec.n_mels, filt, range);
    float *dwin, *dtw, *dfilt; int* drange;
    CKR(dalloc(e->allocs, &dwin, 400)); CKR(dalloc(e->allocs, &dtw, 800));
    CKR(dalloc(e->allocs, &dfilt, filt.size())); CKR(dalloc(e->allocs, &drange, range.size()));
    CKR(upload_sync(dwin, win.data(), 400 * 4));
    CKR(upload_sync(dtw, tw.data(), 800 * 4));
    CKR(upload_sync(dfilt, filt.data(), filt.size() * 4));
    CKR(upload_sync(drange, range.data(), range.size() * 4));
    e->lm.window = dwin; e->lm.twiddle = dtw; e->lm.filters = dfilt; e->lm.frange = drange;
    return WLX_OK;
}

static int load_attn_qkv(Engine* e, WeightSource& ws, const std::string& pre, int d, half_t** W, float** b) {
    int KT;
    CKR(alloc_packed(e, 3 * d, d, W, &KT));
    CKR(pack_into(ws, pre + "q_proj.weight", d, d, *W, KT, 0));
    CKR(pack_into(ws, pre + "k_proj.weight", d, d, *W, KT, d / 16));
    CKR(pack_into(ws, pre + "v_proj.weight", d, d, *W, KT, 2 * d / 16));
    CKR(dalloc(e->allocs, b, (size_t)3 * d));
    CKR(load_vec(ws, pre + "q_proj.bias", d, *b));
    CKR(load_vec(ws, pre + "v_proj.bias", d, *b + 2 * d));   // k_proj has no bias
    return WLX_OK;
}
static int load_linear(Engine* e, WeightSource& ws, const std::string& pre, int N, int K, half_t** W, float** b) {
    int KT;
    CKR(alloc_packed(e, N, K, W, &KT));
    CKR(pack_into(ws, pre + ".weight", N, K, *W, KT, 0));
    CKR(alloc_vec(e, ws, pre + ".bias", N, b));
    return WLX_OK;
}

static int engine_load(Engine* e, const wlx_tensor* weights, int n_weights) {
    const wlx_spec& sp = e->spec;
    const int d = sp.d_model, F = sp.ffn, V = sp.vocab;
    WeightSource ws;
    for (int i = 0; i < n_weights; ++i) ws.by_name[weights[i].name] = &weights[i];
    CK(hipStreamCreateWithFlags(&ws.stream, hipStreamNonBlocking));
    int rc = WLX_OK;
    auto body = [&]() -> int {
        CKR(build_logmel_consts(e));
        // ---- encoder stem
        {
            const wlx_tensor* t; const float* src;
            CKR(need(ws, "model.encoder.conv1.weight", {d, sp.n_mels, 3}, &t));
            e->conv1_KT = (3 * sp.n_mels + 31) / 32;
            CKR(dalloc(e->allocs, &e->conv1_w, (size_t)(d / 16) * e->conv1_KT * 512));
            CKR(ws.device_f32(t, &src));
            launch_pack_conv3(src, d, sp.n_mels, e->conv1_w, e->conv1_KT, ws.stream);
            CKR(need(ws, "model.encoder.conv2.weight", {d, d, 3}, &t));
            CKR(dalloc(e->allocs, &e->conv2_w, (size_t)(d / 16) * (3 * d / 32) * 512));
            CKR(ws.device_f32(t, &src));
            launch_pack_conv3(src, d, d, e->conv2_w, 3 * d / 32, ws.stream);
            CKR(alloc_vec(e, ws, "model.encoder.conv1.bias", d, &e->conv1_b));
            CKR(alloc_vec(e, ws, "model.encoder.conv2.bias", d, &e->conv2_b));
            CKR(need(ws, "model.encoder.embed_positions.weight", {sp.n_audio_ctx, d}, &t));
            CKR(dalloc(e->allocs, &e->enc_pos, (size_t)sp.n_audio_ctx * d));
            CKR(ws.device_f32(t, &src));
            CK(hipMemcpyAsync(e->enc_pos, src, (size_t)sp.n_audio_ctx * d * 4, hipMemcpyDeviceToDevice, ws.stream));
        }
        e->enc.resize(sp.enc_layers);
        for (int l = 0; l < sp.enc_layers; ++l) {
            EncLayerW& w = e->enc[l];
            const std::string p = "model.encoder.layers." + std::to_string(l) + ".";
            CKR(alloc_vec(e, ws, p + "self_attn_layer_norm.weight", d, &w.ln1_g));
            CKR(alloc_vec(e, ws, p + "self_attn_layer_norm.bias", d, &w.ln1_b));
            CKR(load_attn_qkv(e, ws, p + "self_attn.", d, &w.Wqkv, &w.bqkv));
            CKR(load_linear(e, ws, p + "self_attn.out_proj", d, d, &w.Wo, &w.bo));
            CKR(alloc_vec(e, ws, p + "final_layer_norm.weight", d, &w.ln2_g));
            CKR(alloc_vec(e, ws, p + "final_layer_norm.bias", d, &w.ln2_b));
            CKR(load_linear(e, ws, p + "fc1", F, d, &w.W1, &w.b1));
            CKR(load_linear(e, ws, p + "fc2", d, F, &w.W2, &w.b2));
        }
        CKR(alloc_vec(e, ws, "model.encoder.layer_norm.weight", d, &e->enc_ln_g));
        CKR(alloc_vec(e, ws, "model.encoder.layer_norm.bias", d, &e->enc_ln_b));
        // ---- decoder
        {
            const wlx_tensor* t; const float* src;
            CKR(need(ws, "model.decoder.embed_tokens.weight", {V, d}, &t));
            CKR(dalloc(e->allocs, &e->tok_emb16, (size_t)V * d));
            CKR(ws.device_f32(t, &src));
            launch_f32_to_f16(src, e->tok_emb16, (long)V * d, ws.stream);
            int KT;
            CKR(alloc_packed(e, V, d, &e->Wvocab, &KT));
            launch_pack_linear(src, V, d, d, e->Wvocab, KT, 0, ws.stream);   // tied output projection
            CKR(need(ws, "model.decoder.embed_positions.weight", {sp.n_text_ctx, d}, &t));
            CKR(dalloc(e->allocs, &e->dec_pos, (size_t)sp.n_text_ctx * d));
            CKR(ws.device_f32(t, &src));
            CK(hipMemcpyAsync(e->dec_pos, src, (size_t)sp.n_text_ctx * d * 4, hipMemcpyDeviceToDevice, ws.stream));
        }
        e->dec.resize(sp.dec_layers);
        int KTd = d / 32;
        CKR(dalloc(e->allocs, &e->Wckv, (size_t)(sp.dec_layers * 2 * d / 16) * KTd * 512));
        CKR(dalloc(e->allocs, &e->bckv, (size_t)sp.dec_layers * 2 * d));
        for (int l = 0; l < sp.dec_layers; ++l) {
            DecLayerW& w = e->dec[l];
            const std::string p = "model.decoder.layers." + std::to_string(l) + ".";
            CKR(alloc_vec(e, ws, p + "self_attn_layer_norm.weight", d, &w.ln1_g));
            CKR(alloc_vec(e, ws, p + "self_attn_layer_norm.bias", d, &w.ln1_b));
            CKR(load_attn_qkv(e, ws, p + "self_attn.", d, &w.Wqkv, &w.bqkv));
            CKR(load_linear(e, ws, p + "self_attn.out_proj", d, d, &w.Wo, &w.bo));
            CKR(alloc_vec(e, ws, p + "encoder_attn_layer_norm.weight", d, &w.ln2_g));
            CKR(alloc_vec(e, ws, p + "encoder_attn_layer_norm.bias", d, &w.ln2_b));
            CKR(load_linear(e, ws, p + "encoder_attn.q_proj", d, d, &w.Wcq, &w.bcq));
            CKR(load_linear(e, ws, p + "encoder_attn.out_proj", d, d, &w.Wco, &w.bco));
            // cross K/V projections of all layers are fused into one encoder-side GEMM
            CKR(pack_into(ws, p + "encoder_attn.k_proj.weight", d, d, e->Wckv, KTd, l * 2 * d / 16));
            CKR(pack_into(ws, p + "encoder_attn.v_proj.weight", d, d, e->Wckv, KTd, (l * 2 * d + d) / 16));
            CKR(load_vec(ws, p + "encoder_attn.v_proj.bias", d, e->bckv + (size_t)l * 2 * d + d));
            CKR(alloc_vec(e, ws, p + "final_layer_norm.weight", d, &w.ln3_g));
            CKR(alloc_vec(e, ws, p + "final_layer_norm.bias", d, &w.ln3_b));
            CKR(load_linear(e, ws, p + "fc1", F, d, &w.W1, &w.b1));
            CKR(load_linear(e, ws, p + "fc2", d, F, &w.W2, &w.b2));
        }
        CKR(alloc_vec(e, ws, "model.decoder.layer_norm.weight", d, &e->dec_ln_g));
        CKR(alloc_vec(e, ws, "model.decoder.layer_norm.bias", d, &e->dec_ln_b));
        CK(hipStreamSynchronize(ws.stream));
        CK(hipGetLastError());
        return WLX_OK;
    };
    rc = body();
    (void)hipStreamSynchronize(ws.stream);
    if (ws.staging) (void)hipFree(ws.staging);
    (void)hipStreamDestroy(ws.stream);
    return rc;
}

// The process holds device memory that PyTorch (or another runtime) manages — a weight tensor arrived with on_device = 1: the
// embedder issues NULL-stream work of its own (engine.py itself converts weight tensors with torch on the device), and the CU-mask
// constructor only builds BLOCKING streams, which synchronise implicitly with the NULL stream: a second engine loading while a slot
// captures its step graph invalidates the capture (include/wlx.h). With no WLX_SLOT_CU_MASK in the environment such a process gets
// ordinary non-blocking streams.
static std::atomic<bool> g_embedded_device_memory{false};
extern "C" int32_t wlx_engine_create(const wlx_spec* spec, const wlx_tensor* weights, int32_t n_weights,
                                     int32_t device, wlx_engine** out) {
    if (!spec || !weights || !out) return fail(WLX_ERR_ARG, "null argument");
    if (spec->d_model % 64 || spec->d_model / 64 != spec->n_heads)
        return fail(WLX_ERR_ARG, "d_model must be 64*n_heads");
    if (spec->d_model % 128 || spec->d_model > 1536) return fail(WLX_ERR_ARG, "d_model must be a multiple of 128, <= 1536");
    if (spec->ffn % 128) return fail(WLX_ERR_ARG, "ffn must be a multiple of 128");
    if (spec->n_mels != 80 && spec->n_mels != 128) return fail(WLX_ERR_ARG, "n_mels must be 80 or 128");
    if (spec->n_audio_ctx != WLX_T_AUDIO || spec->n_text_ctx != WLX_T_TEXT)
        return fail(WLX_ERR_ARG, "n_audio_ctx/n_text_ctx must be 1500/448");
    if (spec->vocab > 1024 * 52) return fail(WLX_ERR_ARG, "vocab too large");
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(WLX_ERR_ARG, "device %d out of range (%d)", device, ndev);
    CK(hipSetDevice(device));
    if (int pe = gemm_prepare_device()) return fail(WLX_ERR_HIP, "GEMM kernel set-up failed on device %d (hip error %d)", device, pe);
    wlx_engine* e = new wlx_engine();
    e->spec = *spec;
    e->device = device;
    e->H = spec->n_heads;
    for (int i = 0; i < n_weights; ++i)
        if (weights[i].on_device) { g_embedded_device_memory.store(true); break; }
    const char* ng = getenv("WLX_NO_GRAPH");
    e->use_graph = !(ng && ng[0] == '1');
    const char* v1 = wlx_ab("WLX_DECODE_V1");
    g_decode_v1 = (v1 && v1[0] == '1');
    int rc = engine_load(e, weights, n_weights);
    if (rc != WLX_OK) {
        for (void* p : e->allocs) (void)hipFree(p);
        delete e;
        return rc;
    }
    *out = e;
    return WLX_OK;
}

// live slots per device that may hold a hardware queue of their own (WLX_DEDICATED_QUEUES, default 4 = the reference server's
// max_clients); the SAME number is the demotion threshold: one more live slot and every slot goes back to the shared pool
static int max_dedicated_queues() {
    static const int v = [] { const char* e = getenv("WLX_DEDICATED_QUEUES"); return e ? atoi(e) : 4; }();
    return v;
}
static std::atomic<int> g_dedicated_live[64];      // live slots with a hardware queue of their own, per device (create_slot_stream)
static std::atomic<int> g_slots_live[64];          // live slots per device
static std::atomic<bool> g_demote[64];             // more live slots than dedicated queues allowed: dedicated slots fall back at their next call
static std::atomic<unsigned> g_promote_epoch[64];  // raised when g_demote falls: a shared-pool slot tries ONCE per epoch to take a queue of its own
static void slot_free(Slot* s) {
    if (!s) return;
    if (s->device_of >= 0 && s->device_of < 64) {
        if (s->dedicated_queue) g_dedicated_live[s->device_of].fetch_sub(1);
        if (s->counted && g_slots_live[s->device_of].fetch_sub(1) - 1 <= max_dedicated_queues()) {
            if (g_demote[s->device_of].exchange(false)) g_promote_epoch[s->device_of].fetch_add(1);
        }
    }
    if (s->align_scores) (void)hipFree(s->align_scores);
    for (auto& kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    if (s->h_gen) (void)hipHostFree(s->h_gen);
    if (s->h_hyp) (void)hipHostFree(s->h_hyp);
    if (s->h_pf) (void)hipHostFree(s->h_pf);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->ev_lm0) (void)hipEventDestroy(s->ev_lm0);
    if (s->ev_lm1) (void)hipEventDestroy(s->ev_lm1);
    if (s->ev_en0) (void)hipEventDestroy(s->ev_en0);
    if (s->ev_en1) (void)hipEventDestroy(s->ev_en1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

extern "C" void wlx_engine_destroy(wlx_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (Slot* s : e->slots) slot_free(s);
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
}

extern "C" int32_t wlx_engine_spec(const wlx_engine* e, wlx_spec* out) {
    if (!e || !out) return fail(WLX_ERR_ARG, "null argument");
    *out = e->spec;
    return WLX_OK;
}

// ------------------------------------------------------------------------------------------------
// slots
// Every entry point holds the slot's call mutex for its duration: a slot is one unit of concurrency (one stream, one set
// of scratch buffers), so a second call on it is refused rather than corrupting the first, and wlx_slot_destroy waits for
// a call in flight instead of freeing buffers under it.
struct SlotGuard {
    Slot* s = nullptr;
    ~SlotGuard() { if (s) s->call_mu.unlock(); }
};
static int create_slot_stream(int device, hipStream_t* out, bool* dedicated_out);
static bool dedicated_streams_possible();
static int slot_acquire(wlx_engine* e, int slot, SlotGuard& g) {
    if (!e) return fail(WLX_ERR_ARG, "null engine");
    Slot* s = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (slot < 0 || slot >= (int)e->slots.size() || !e->slots[slot]) return fail(WLX_ERR_ARG, "bad slot %d", slot);
        s = e->slots[slot];
        if (!s->call_mu.try_lock())
            return fail(WLX_ERR_STATE, "slot %d is busy in another call (a slot serves one call at a time)", slot);
        g.s = s;
    }
    // From here on the slot is ours (call_mu) and the engine mutex is released: a stream swap below waits for the slot's own
    // stream, which must not keep other slots out of the library.
    const int dv = s->device_of;
    if (dv < 0 || dv >= 64) return WLX_OK;
    // more slots than dedicated hardware queues exist on this device now: give the queue back (see create_slot_stream: past
    // ~6 busy hardware queues everything collapses; eight ordinary streams over the shared pool run at 1883 xRT, eight with
    // four dedicated queues at 1145). The slot's captured graphs do not depend on the stream they were captured on.
    // ... or the process has since become an embedder of another runtime's device memory (a later engine's weights arrived as device
    // pointers: g_embedded_device_memory): the CU-mask streams are BLOCKING streams and that runtime's NULL-stream work would invalidate
    // this slot's captures, so a slot created before that engine gives its stream back at its next call too (the flag was
    // read at stream creation only).
    if (s->dedicated_queue && (g_demote[dv].load() || !dedicated_streams_possible())) {
        hipStream_t ns = nullptr;
        if (hipSetDevice(dv) == hipSuccess && hipStreamSynchronize(s->stream) == hipSuccess &&
            hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) == hipSuccess) {
            (void)hipStreamDestroy(s->stream);
            s->stream = ns;
            s->dedicated_queue = false;
            g_dedicated_live[dv].fetch_sub(1);
        } else {
            (void)hipGetLastError();
        }
    } else if (!s->dedicated_queue && s->counted && !g_demote[dv].load() && dedicated_streams_possible() &&
               s->promote_epoch != g_promote_epoch[dv].load() && g_dedicated_live[dv].load() < max_dedicated_queues()) {
        // ... and back: the device is down to <= WLX_DEDICATED_QUEUES live slots again (the fifth client left, a batch
        // lane was released) — a slot that was demoted, or created while the device was crowded, takes a queue of its own at its
        // next call. ONE attempt per slot and per "crowd left" event (g_promote_epoch, raised where g_demote falls): with
        // WLX_SLOT_CU_MASK=off, or where the dedicated constructor fails, the steady state costs nothing — the unconditional form
        // synchronised the slot stream and created + destroyed a stream on EVERY call.
        s->promote_epoch = g_promote_epoch[dv].load();
        hipStream_t ns = nullptr;
        bool ded = false;
        if (hipSetDevice(dv) == hipSuccess && hipStreamSynchronize(s->stream) == hipSuccess &&
            create_slot_stream(dv, &ns, &ded) == WLX_OK) {
            if (ded) { (void)hipStreamDestroy(s->stream); s->stream = ns; s->dedicated_queue = true; }
            else (void)hipStreamDestroy(ns);
        } else {
            (void)hipGetLastError();
        }
    }
    return WLX_OK;
}

static int slot_grow_audio(Engine* e, Slot* s, size_t n_samples) {
    if (n_samples <= s->pcm_cap) return WLX_OK;
    // grow PCM + feature buffers (sizes rounded up to whole 30 s windows)
    size_t cap = ((n_samples + 479999) / 480000) * 480000;
    CK(hipStreamSynchronize(s->stream));
    float* npcm; float* nfe;
    long ld = (long)(cap / 160 + 64);
    CKR(dalloc(s->allocs, &npcm, (size_t)s->B * cap, false));
    CKR(dalloc(s->allocs, &nfe, (size_t)s->B * e->spec.n_mels * ld));
    // (old buffers stay in the pool until slot destruction; growth is rare: 1-2 times per stream)
    if (s->pcm_cap) {
        // keep what the other items of a batch already hold: a batched encode runs log-mel item by item, and a later,
        // longer item must not wipe the earlier items' PCM / features / frame counts
        CK(hipMemcpy2DAsync(npcm, cap * sizeof(float), s->pcm, s->pcm_cap * sizeof(float), s->pcm_cap * sizeof(float),
                            (size_t)s->B, hipMemcpyDeviceToDevice, s->stream));
        CK(hipMemcpy2DAsync(nfe, (size_t)ld * sizeof(float), s->feats, (size_t)s->feat_ld * sizeof(float),
                            (size_t)s->feat_ld * sizeof(float), (size_t)s->B * e->spec.n_mels, hipMemcpyDeviceToDevice, s->stream));
        CK(hipStreamSynchronize(s->stream));
    }
    s->pcm = npcm; s->pcm_cap = cap; s->feats = nfe; s->feat_ld = ld;
    return WLX_OK;
}

// A slot's stream gets a HARDWARE QUEUE OF ITS OWN. HIP multiplexes ordinary streams round robin over GPU_MAX_HW_QUEUES = 4
// hardware queues (the null stream and this library's utility stream take two of them), so with four concurrent clients two
// slots shared one queue and their decode chains serialised: 4 streams ran at 1826 xRT. A stream created through
// hipExtStreamCreateWithCUMask owns its queue (the CU mask is a property of the queue) — here with EVERY CU enabled, so
// nothing is restricted: 2734 xRT for the same four streams, single stream unchanged (profiles/r3b_streams4_*.json,
// profiles/r3b_streams4_dedicated_queues_overlap.txt). Raising GPU_MAX_HW_QUEUES instead is NOT an option: 5, 6 and 8 run the
// same workload at 610-640 xRT. WLX_SLOT_CU_MASK=off restores ordinary streams; =stride4 / contig4 give slot k a quarter of
// the CUs (every 4th CU: a decode step is as fast on 64 CUs spread over all XCDs as on 256 — 373 us — but the encoder is not).
// Only the first WLX_DEDICATED_QUEUES (4 = the reference server's max_clients) live slots of a device get one: with 8 dedicated
// queues (+ the null and utility streams' two) the same 8-stream workload drops from 1883 xRT (shared queues) to 1135 — past
// some number of busy hardware queues the command processor time-slices them (profiles/r3c_bench_s8_default.json).
static std::string slot_stream_mode() {
    static const char* cu_mode = getenv("WLX_SLOT_CU_MASK");
    if (cu_mode) return cu_mode;
    return g_embedded_device_memory.load() ? "off" : "full";
}
static bool dedicated_streams_possible() { return slot_stream_mode() != "off" && max_dedicated_queues() > 0; }
static int create_slot_stream(int device, hipStream_t* out, bool* dedicated_out) {
    const int max_dedicated = max_dedicated_queues();
    static std::atomic<int> slot_seq{0};
    std::string m = slot_stream_mode();
    {   // once per process AND MODE: which kind of stream the slots get (the default synchronises implicitly with the NULL
        // stream; r05: the mode can change when an engine brings device-resident weights — say so again then)
        static std::atomic<int> said{0};
        const int bit = (m == "off") ? 1 : 2;
        if (!(said.fetch_or(bit) & bit) && getenv("WLX_QUIET") == nullptr)
            fprintf(stderr, "[wlx] slot streams: WLX_SLOT_CU_MASK=%s%s, up to %d hardware queues per device%s\n", m.c_str(),
                    (m == "off" && g_embedded_device_memory.load()) ? " (weights were handed over as device pointers of another runtime: non-blocking streams unless the variable says otherwise)" : "",
                    max_dedicated,
                    m == "off" ? "" : " (CU-mask streams are blocking streams: keep NULL-stream work of this process off the device, or set WLX_SLOT_CU_MASK=off)");
    }
    *dedicated_out = false;
    if (m != "off") {
        if (device < 0 || device >= 64 || g_dedicated_live[device].fetch_add(1) >= max_dedicated) {
            if (device >= 0 && device < 64) g_dedicated_live[device].fetch_sub(1);
            m = "off";
        } else {
            *dedicated_out = true;
        }
    }
    auto undo = [&] { if (*dedicated_out) { g_dedicated_live[device].fetch_sub(1); *dedicated_out = false; } };
    if (m == "prio_high" || m == "prio_alt") {
        // (A/B) priority streams draw their hardware queues from a per-priority pool, separate from the normal-priority pool
        // the null and utility streams live in, and — unlike the CU-mask constructor — take the non-blocking flag
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));          // numerically lower = higher priority
        const int k = slot_seq.fetch_add(1);
        const int pr = (m == "prio_alt" && (k & 1)) ? lo : hi;
        if (hipStreamCreateWithPriority(out, hipStreamNonBlocking, pr) == hipSuccess) return WLX_OK;
        (void)hipGetLastError();
        undo();
    } else if (m != "off") {
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        const int k = slot_seq.fetch_add(1);
        std::vector<uint32_t> mask(words, 0u);
        const int parts = (m == "contig2") ? 2 : ((m == "stride4" || m == "contig4") ? 4 : 1);
        for (int cu = 0; cu < ncu; ++cu) {
            const bool mine = (m == "stride4") ? (cu % parts == k % parts) : (cu * parts / ncu == k % parts);
            if (mine) mask[cu >> 5] |= 1u << (cu & 31);
        }
        if (hipExtStreamCreateWithCUMask(out, (uint32_t)words, mask.data()) == hipSuccess) return WLX_OK;
        (void)hipGetLastError();           // not supported here: an ordinary stream (shared queues) is still correct
        undo();
    }
    CK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return WLX_OK;
}

extern "C" int32_t wlx_slot_create(wlx_engine* e, int32_t max_batch, int32_t max_rows_per_item, int32_t* slot_out) {
    if (!e || !slot_out) return fail(WLX_ERR_ARG, "null argument");
    if (max_batch < 1 || max_batch > 64) return fail(WLX_ERR_ARG, "max_batch out of range");
    if (max_rows_per_item < 1 || max_rows_per_item > 16) return fail(WLX_ERR_ARG, "max_rows_per_item must be 1..16");
    if (max_batch * max_rows_per_item > WLX_MAX_DEC_ROWS) return fail(WLX_ERR_ARG, "max_batch*max_rows_per_item must be <= %d", WLX_MAX_DEC_ROWS);
    CK(hipSetDevice(e->device));
    const wlx_spec& sp = e->spec;
    const int d = sp.d_model, F = sp.ffn, L = sp.dec_layers, B = max_batch, R = max_rows_per_item;
    {   // what the slot will hold (its large terms), checked against the device BEFORE the first allocation (--batch_max_size 64
        // builds 64 x 5-row slots — ~11 GB per lane for Whisper-small, ~45 GB for large-v3 — and used to fail, if at all, deep inside
        // the allocation list with a bare hipMalloc error)
        const double TBd = (double)B * WLX_T_AUDIO, rows = (double)std::max(64, B * R);
        const double bytes = 2.0 * B * 480000 * 4 + (double)B * ((WLX_N_FRAMES + 2) * (sp.n_mels * 2.0 + d * 2.0)) +       // PCM + features, window, conv
                             TBd * d * (4 + 2 + 2 + 2 + 2 + 4) + 2.0 * B * WLX_T_AUDIO_PAD * d * 2 + TBd * F * 2 +            // encoder activations
                             2.0 * L * B * WLX_T_AUDIO_PAD * d * 2 +                                                           // cross K / V
                             2.0 * L * (double)B * R * WLX_T_TEXT * d * 2 +                                                     // self-attention KV cache
                             rows * (((sp.vocab + 15) / 16) * 16 * 4.0 + d * 8.0 + F * 2.0) + (double)WLX_T_TEXT * (d * 16.0 + F * 2.0);   // logits, step + prefill rows
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if (bytes > (double)free_b)
                return fail(WLX_ERR_NOMEM, "a slot of %d items x %d rows needs ~%.1f GB of device memory (encoder activations, cross K/V, KV cache, logits); "
                            "%.1f GB are free on device %d: lower max_batch (--batch_max_size) or the number of lanes", B, R, bytes / 1e9, free_b / 1e9, e->device);
        } else (void)hipGetLastError();
        if (bytes > 8e9 && getenv("WLX_QUIET") == nullptr)
            fprintf(stderr, "[wlx] slot of %d items x %d rows: ~%.1f GB of device memory\n", B, R, bytes / 1e9);
    }
    Slot* s = new Slot();
    s->B = B; s->R = R; s->cache_rows = B * R; s->rows_cap = std::max(64, B * R); s->groups_cap = std::max(B, 4);
    s->nframes.assign(B, 0);
    s->npcm.assign(B, 0);
    int rc = [&]() -> int {
        s->device_of = e->device;
        if (e->device >= 0 && e->device < 64) {
            s->counted = true;
            s->promote_epoch = g_promote_epoch[e->device].load();
            if (g_slots_live[e->device].fetch_add(1) + 1 > max_dedicated_queues()) g_demote[e->device].store(true);
        }
        if (s->counted && g_demote[e->device].load()) {
            CK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));        // past 4 live slots: the shared pool
        } else {
            CKR(create_slot_stream(e->device, &s->stream, &s->dedicated_queue));
        }
        CK(hipEventCreate(&s->ev0)); CK(hipEventCreate(&s->ev1));
        CK(hipEventCreate(&s->ev_lm0)); CK(hipEventCreate(&s->ev_lm1));
        CK(hipEventCreate(&s->ev_en0)); CK(hipEventCreate(&s->ev_en1));
        CKR(slot_grow_audio(e, s, 480000));
        CKR(dalloc(s->allocs, &s->gmax, (size_t)B));
        CKR(dalloc(s->allocs, &s->d_rng, (size_t)2 * WLX_LM_MAXRANGES));
        s->featT_stride = (long)(WLX_N_FRAMES + 2) * sp.n_mels + 64;
        s->h1_stride = (long)(WLX_N_FRAMES + 2) * d;
        CKR(dalloc(s->allocs, &s->featT, (size_t)B * s->featT_stride));
        CKR(dalloc(s->allocs, &s->h1, (size_t)B * s->h1_stride));
        const size_t TB = (size_t)B * WLX_T_AUDIO;
        CKR(dalloc(s->allocs, &s->x, TB * d));
        CKR(dalloc(s->allocs, &s->ln, TB * d));
        CKR(dalloc(s->allocs, &s->q, TB * d));
        CKR(dalloc(s->allocs, &s->k, (size_t)B * WLX_T_AUDIO_PAD * d));
        CKR(dalloc(s->allocs, &s->vt, (size_t)B * d * WLX_T_AUDIO_PAD));
        CKR(dalloc(s->allocs, &s->attn, TB * d));
        CKR(dalloc(s->allocs, &s->h2, TB * F));
        CKR(dalloc(s->allocs, &s->enc16, TB * d));
        CKR(dalloc(s->allocs, &s->enc32, TB * d));
        CKR(dalloc(s->allocs, &s->ck, (size_t)L * B * WLX_T_AUDIO_PAD * d));
        CKR(dalloc(s->allocs, &s->cvt, (size_t)L * B * d * WLX_T_AUDIO_PAD));
        CKR(dalloc(s->allocs, &s->kc, (size_t)L * s->cache_rows * WLX_T_TEXT * d));
        CKR(dalloc(s->allocs, &s->vc, (size_t)L * s->cache_rows * WLX_T_TEXT * d));
        const int RC = s->rows_cap;
        CKR(dalloc(s->allocs, &s->xd, (size_t)RC * d));
        CKR(dalloc(s->allocs, &s->qd, (size_t)RC * d));
        CKR(dalloc(s->allocs, &s->attnd, (size_t)RC * d));
        CKR(dalloc(s->allocs, &s->hd, (size_t)RC * F));
        CKR(dalloc(s->allocs, &s->slab, (size_t)WLX_FC2_KS * RC * d));
        CKR(dalloc(s->allocs, &s->part_o, (size_t)s->groups_cap * e->H * WLX_XSPLIT * 16 * 64));
        CKR(dalloc(s->allocs, &s->part_ml, (size_t)s->groups_cap * e->H * WLX_XSPLIT * 16 * 2));
        {   // the one-pass prompt prefill's working set: up to WLX_T_TEXT rows (engine.hip prefill_tokens)
            const size_t PR = WLX_T_TEXT, PG = (WLX_T_TEXT + 15) / 16;
            Slot::DecBufs& b = s->pf;
            b.slab_rows = (int)PR;
            CKR(dalloc(s->allocs, &b.xd, PR * d)); CKR(dalloc(s->allocs, &b.qd, PR * d)); CKR(dalloc(s->allocs, &b.attnd, PR * d));
            CKR(dalloc(s->allocs, &b.hd, PR * F)); CKR(dalloc(s->allocs, &b.slab, (size_t)WLX_FC2_KS * PR * d));
            CKR(dalloc(s->allocs, &b.part_o, PG * e->H * WLX_XSPLIT * 16 * 64));
            CKR(dalloc(s->allocs, &b.part_ml, PG * e->H * WLX_XSPLIT * 16 * 2));
            CKR(dalloc(s->allocs, &b.d_token, PR)); CKR(dalloc(s->allocs, &b.d_pos, PR)); CKR(dalloc(s->allocs, &b.d_cache, PR));
            CKR(dalloc(s->allocs, &b.d_ancrow, PR)); CKR(dalloc(s->allocs, &b.d_group_item, PG));
            // usable when every projection of this model takes the lean kernel in row chunks (d_model a multiple of 256, ...)
            GemvParams q{};
            static const float dummy_bias = 0.f;
            q.M = 96; q.bias = &dummy_bias; q.xsrc = GEMV_X_PLAIN;
            auto lean = [&](int in, int out, int N, int K) { q.in_mode = in; q.out_mode = out; q.N = N; q.K = K; q.KT = K / 32; return dec_gemv_is_lean(q); };
            s->pf_ok = lean(GEMV_IN_LN, GEMV_OUT_QKV, 3 * d, d) && lean(GEMV_IN_F16, GEMV_OUT_RESID, d, d) && lean(GEMV_IN_LN, GEMV_OUT_F16, d, d) &&
                       lean(GEMV_IN_LN, GEMV_OUT_GELU_F16, F, d) && lean(GEMV_IN_F16, GEMV_OUT_RESID, d, F);
        }
        s->ldl = ((sp.vocab + 15) / 16) * 16;
        CKR(dalloc(s->allocs, &s->logits, (size_t)RC * s->ldl));
        CKR(dalloc(s->allocs, &s->d_token, (size_t)RC));
        CKR(dalloc(s->allocs, &s->d_pos, (size_t)RC));
        CKR(dalloc(s->allocs, &s->d_cache, (size_t)RC));
        CKR(dalloc(s->allocs, &s->d_ancrow, (size_t)RC));
        CKR(dalloc(s->allocs, &s->d_group_item, (size_t)RC));
        const int CR = std::max(s->cache_rows, RC);
        CKR(dalloc(s->allocs, &s->d_anc, (size_t)CR * WLX_T_TEXT));
        CKR(dalloc(s->allocs, &s->d_intok, (size_t)CR * WLX_T_TEXT));
        SearchState& st = s->st;
        CKR(dalloc(s->allocs, &st.step, 1)); CKR(dalloc(s->allocs, &st.done, 1)); CKR(dalloc(s->allocs, &st.n_finished, 1));
        CKR(dalloc(s->allocs, &st.item_done, (size_t)B)); CKR(dalloc(s->allocs, &st.plen, (size_t)B));
        CKR(dalloc(s->allocs, &st.cum, (size_t)RC)); CKR(dalloc(s->allocs, &st.row_done, (size_t)RC));
        CKR(dalloc(s->allocs, &st.cand_score, (size_t)RC * WLX_MAX_CAND));
        CKR(dalloc(s->allocs, &st.cand_tok, (size_t)RC * WLX_MAX_CAND));
        CKR(dalloc(s->allocs, &st.samp_tok, (size_t)RC)); CKR(dalloc(s->allocs, &st.samp_lp, (size_t)RC));
        // the results of a generate live in pinned host memory (round 6, search.hip finish_item): the update kernels store them there and the host
        // reads them when it sees the done word — no copies, no wait for the step that is already in flight behind the finish
        {
            const size_t ints = (size_t)B * (2 + 2 * WLX_MAX_HYP + (size_t)WLX_MAX_HYP * WLX_T_TEXT);
            CK(hipHostMalloc(reinterpret_cast<void**>(&s->h_hyp), ints * sizeof(int), hipHostMallocDefault));
            memset(s->h_hyp, 0, ints * sizeof(int));
            s->max_items = B;
            CK(hipHostMalloc(reinterpret_cast<void**>(&s->h_pf), (size_t)(5 * WLX_T_TEXT + 64) * sizeof(int), hipHostMallocDefault));
            st.n_hyp_host = s->h_hyp;
            st.hyp_len = st.n_hyp_host + B;
            st.hyp_score = reinterpret_cast<float*>(st.hyp_len + (size_t)B * WLX_MAX_HYP);
            st.no_speech = st.hyp_score + (size_t)B * WLX_MAX_HYP;
            st.hyp_tokens = reinterpret_cast<int*>(st.no_speech + B);
        }
        CKR(dalloc(s->allocs, &st.n_hyp, (size_t)B));
        CKR(dalloc(s->allocs, &st.nsp_row, (size_t)RC));
        st.token = s->d_token; st.pos = s->d_pos; st.anc = s->d_anc; st.intok = s->d_intok;
        CKR(dalloc(s->allocs, &st.scan_stats, (size_t)RC * SC_MAXCH * SC_NSTAT));
        CKR(dalloc(s->allocs, &st.scan_cv, (size_t)RC * (SC_MAXCH + 1) * WLX_MAX_CAND));
        CKR(dalloc(s->allocs, &st.scan_ci, (size_t)RC * (SC_MAXCH + 1) * WLX_MAX_CAND));
        CKR(dalloc(s->allocs, &st.rule, (size_t)RC * 4));
        CKR(dalloc(s->allocs, &s->d_align_tgt, (size_t)WLX_T_TEXT));
        CKR(dalloc(s->allocs, &s->d_align_prob, (size_t)WLX_T_TEXT));
        CKR(dalloc(s->allocs, &s->d_sp, 1));
        CKR(dalloc(s->allocs, &s->d_suppress, (size_t)(1024 * 52 / 32)));
        CKR(dalloc(s->allocs, &s->d_lang_ids, 256));
        CKR(dalloc(s->allocs, &s->d_probs, (size_t)B * 256));
        CKR(dalloc(s->allocs, &s->d_tokprob, (size_t)RC));
        s->h_stage_ints = 1 << 16;
        CK(hipHostMalloc(reinterpret_cast<void**>(&s->h_stage), s->h_stage_ints * sizeof(int), hipHostMallocDefault));
        // the search kernels raise this pinned word themselves when every item is finished (no per-step D2H copy)
        s->st.done_host = s->h_stage + (s->h_stage_ints - 4);
        // wlx_generate's pinned staging: [set-up: SearchParams | cum | rule | plen | nsp | ancestry rows] [results: n_hyp |
        // hyp_len | hyp_score | no_speech | step | hyp_tokens]
        s->h_gen_bytes = 4096 + (size_t)RC * (4 + 16 + 4) + (size_t)B * 4 + (size_t)RC * WLX_T_TEXT * 2 + 256 + ((size_t)RC * 4 + B) * 4 + 128 +
                         (size_t)B * (4 + WLX_MAX_HYP * 8 + 4) + 64 + (size_t)B * WLX_MAX_HYP * WLX_T_TEXT * 4;
        CK(hipHostMalloc(reinterpret_cast<void**>(&s->h_gen), s->h_gen_bytes, hipHostMallocDefault));
        CK(hipStreamSynchronize(s->stream));      // (allocations were zeroed on the utility stream and waited for in dalloc)
        return WLX_OK;
    }();
    if (rc != WLX_OK) { slot_free(s); return rc; }
    std::lock_guard<std::mutex> g(e->mu);
    // ids are never reused: a stale handle of a destroyed slot must fail ("bad slot"), not reach its successor
    e->slots.push_back(s);
    const int id = (int)e->slots.size() - 1;
    *slot_out = id;
    return WLX_OK;
}

extern "C" int32_t wlx_slot_destroy(wlx_engine* e, int32_t slot) {
    if (!e) return fail(WLX_ERR_ARG, "null engine");
    Slot* s = nullptr;
    for (;;) {      // wait for a call still running on the slot in another thread (a session being torn down mid-chunk)
        {
            std::lock_guard<std::mutex> g(e->mu);
            if (slot < 0 || slot >= (int)e->slots.size() || !e->slots[slot]) return fail(WLX_ERR_ARG, "bad slot %d", slot);
            if (e->slots[slot]->call_mu.try_lock()) {
                s = e->slots[slot];
                e->slots[slot] = nullptr;        // no other thread can reach it any more
                break;
            }
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    s->call_mu.unlock();
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(s->stream);
    slot_free(s);
    return WLX_OK;
}

// Log-mel requests are collected and launched together (round 4): wlx_logmel_resident only records the item; the launches — ONE set of
// three kernels for all recorded items — go out in front of the first consumer of the features (wlx_encode, wlx_features_get), of a
// timing read, or of anything that overwrites what a recorded item depends on (its PCM, the audio buffers).
static int flush_logmel(Engine* e, Slot* s) {
    if (s->lm_items.empty()) return WLX_OK;
    CK(hipSetDevice(e->device));
    CK(hipEventRecord(s->ev_lm0, s->stream));
    for (size_t i0 = 0; i0 < s->lm_items.size(); i0 += WLX_LM_MAXB) {
        LogmelBatch lb{};
        for (size_t i = i0; i < s->lm_items.size() && i < i0 + WLX_LM_MAXB; ++i) {
            const int item = s->lm_items[i], k = lb.n_items++;
            const int64_t n = s->npcm[item];
            lb.pcm[k] = s->pcm + (size_t)item * s->pcm_cap; lb.n[k] = (long)n;
            lb.feats[k] = s->feats + (size_t)item * e->spec.n_mels * s->feat_ld; lb.T[k] = (int)((n + 160) / 160);
            lb.gmax[k] = s->gmax + item;
        }
        launch_logmel_batch(lb, e->spec.n_mels, e->lm, s->feat_ld, s->stream);
    }
    s->lm_items.clear();
    CK(hipGetLastError());
    CK(hipEventRecord(s->ev_lm1, s->stream));
    s->lm_pending = true;
    return WLX_OK;
}

extern "C" int32_t wlx_sync(wlx_engine* e, int32_t slot) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    CK(hipSetDevice(e->device));
    CKR(flush_logmel(e, s));
    CK(hipStreamSynchronize(s->stream));
    return WLX_OK;
}

extern "C" int32_t wlx_timings_get(wlx_engine* e, int32_t slot, wlx_timings* out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (!out) return fail(WLX_ERR_ARG, "null out");
    CKR(flush_logmel(e, s));
    if (s->en_pending) {                      // the last encoder pass
        CK(hipSetDevice(e->device));
        CK(hipEventSynchronize(s->ev_en1));
        CK(hipEventElapsedTime(&s->tm.encode_ms, s->ev_en0, s->ev_en1));
        s->en_pending = false;
    }
    if (s->gen_pending) {                     // the last generate: its results were read from pinned memory without waiting for the stream
        CK(hipSetDevice(e->device));
        CK(hipEventSynchronize(s->ev1));
        CK(hipEventElapsedTime(&s->tm.generate_ms, s->ev0, s->ev1));
        int steps = 0;
        CK(hipMemcpy(&steps, s->st.step, 4, hipMemcpyDeviceToHost));
        s->tm.decode_steps = steps;
        s->gen_pending = false;
    }
    if (s->lm_pending) {                      // the last log-mel launch: waited for here, not in wlx_logmel_resident
        CK(hipSetDevice(e->device));
        CK(hipEventSynchronize(s->ev_lm1));
        CK(hipEventElapsedTime(&s->tm.logmel_ms, s->ev_lm0, s->ev_lm1));
        s->lm_pending = false;
    }
    *out = s->tm;
    return WLX_OK;
}

// ------------------------------------------------------------------------------------------------
// log-mel
extern "C" int32_t wlx_pcm_put(wlx_engine* e, int32_t slot, int32_t item, const float* pcm, int64_t n) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (!pcm || n <= 0) return fail(WLX_ERR_ARG, "empty audio");
    if (item < 0 || item >= s->B) return fail(WLX_ERR_ARG, "bad item %d", item);
    if (n > 16000LL * 3600) return fail(WLX_ERR_ARG, "audio chunk too long");
    CK(hipSetDevice(e->device));
    if ((size_t)n > s->pcm_cap || std::find(s->lm_items.begin(), s->lm_items.end(), (int)item) != s->lm_items.end())
        CKR(flush_logmel(e, s));            // a recorded log-mel request reads this item's PCM (or the buffers are about to be re-allocated)
    CKR(slot_grow_audio(e, s, (size_t)n));
    float* dp = s->pcm + (size_t)item * s->pcm_cap;
    CK(hipMemcpyAsync(dp, pcm, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s->stream));
    CK(hipStreamSynchronize(s->stream));   // the caller's PCM buffer may be reused after return
    s->npcm[item] = n;
    return WLX_OK;
}

// ------------------------------------------------------------------------------------------------
// PCM ring (include/wlx.h): whisper_live/backend/base.py:173-234 on the device
extern "C" int32_t wlx_ring_create(wlx_engine* e, int64_t capacity_samples, wlx_ring** out) {
    if (!e || !out) return fail(WLX_ERR_ARG, "null argument");
    if (capacity_samples < 0 || capacity_samples > 16000LL * 3600) return fail(WLX_ERR_ARG, "ring capacity out of range");
    CK(hipSetDevice(e->device));
    wlx_ring* r = new wlx_ring();
    r->device = e->device;
    r->cap = (size_t)(capacity_samples > 0 ? capacity_samples : 16000LL * 64);
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&r->buf), r->cap * sizeof(float));
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipEventCreateWithFlags(&r->last_read, hipEventDisableTiming);
    if (he != hipSuccess) {
        (void)hipGetLastError();
        wlx_ring_destroy(r);
        return fail(WLX_ERR_HIP, "wlx_ring_create: %s", hipGetErrorString(he));
    }
    *out = r;
    return WLX_OK;
}

extern "C" void wlx_ring_destroy(wlx_ring* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->read_pending && r->last_read) (void)hipEventSynchronize(r->last_read);
    if (r->stream) { (void)hipStreamSynchronize(r->stream); (void)hipStreamDestroy(r->stream); }
    if (r->last_read) (void)hipEventDestroy(r->last_read);
    if (r->buf) (void)hipFree(r->buf);
    delete r;
}

// readers launched but possibly not run yet: wait for them before data moves under them
static int ring_wait_readers(Ring* r) {
    if (r->read_pending) { CK(hipEventSynchronize(r->last_read)); r->read_pending = false; }
    return WLX_OK;
}

extern "C" int32_t wlx_ring_append(wlx_ring* r, const float* samples, int64_t n, int64_t max_resident, int64_t trim,
                                   int64_t* dropped_out, int64_t* base_out, int64_t* resident_out) {
    if (!r || n < 0 || (n > 0 && !samples)) return fail(WLX_ERR_ARG, "wlx_ring_append: bad argument");
    if (n > 16000LL * 3600) return fail(WLX_ERR_ARG, "audio chunk too long");
    std::lock_guard<std::mutex> lk(r->mu);
    CK(hipSetDevice(r->device));
    int64_t dropped = 0;
    if (max_resident > 0 && trim > 0 && r->resident > max_resident) {
        // add_frames (base.py:191-198): the buffer is past its cap — the oldest `trim` samples go, the rest moves to the front
        dropped = std::min<int64_t>(trim, r->resident);
        const int64_t keep = r->resident - dropped;
        CKR(ring_wait_readers(r));
        for (int64_t done = 0; done < keep; done += dropped) {                      // pieces of <= `dropped` samples: source and destination never overlap
            const int64_t m = std::min<int64_t>(dropped, keep - done);
            CK(hipMemcpyAsync(r->buf + done, r->buf + dropped + done, (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, r->stream));
        }
        r->base += dropped;
        r->resident = keep;
    }
    if ((size_t)(r->resident + n) > r->cap) {                                      // a packet larger than the head-room: grow (rare)
        size_t cap = r->cap;
        while (cap < (size_t)(r->resident + n)) cap *= 2;
        float* nb = nullptr;
        CK(hipMalloc(reinterpret_cast<void**>(&nb), cap * sizeof(float)));
        CKR(ring_wait_readers(r));
        CK(hipMemcpyAsync(nb, r->buf, (size_t)r->resident * sizeof(float), hipMemcpyDeviceToDevice, r->stream));
        CK(hipStreamSynchronize(r->stream));
        CK(hipFree(r->buf));
        r->buf = nb; r->cap = cap;
    }
    if (n > 0) CK(hipMemcpyAsync(r->buf + r->resident, samples, (size_t)n * sizeof(float), hipMemcpyHostToDevice, r->stream));
    CK(hipStreamSynchronize(r->stream));            // the caller's buffer may be reused; readers launched after this return see the samples
    r->resident += n;
    if (dropped_out) *dropped_out = dropped;
    if (base_out) *base_out = r->base;
    if (resident_out) *resident_out = r->resident;
    return WLX_OK;
}

extern "C" int32_t wlx_ring_state(wlx_ring* r, int64_t* base_out, int64_t* resident_out) {
    if (!r) return fail(WLX_ERR_ARG, "null ring");
    std::lock_guard<std::mutex> lk(r->mu);
    if (base_out) *base_out = r->base;
    if (resident_out) *resident_out = r->resident;
    return WLX_OK;
}

extern "C" int32_t wlx_logmel_ring(wlx_engine* e, int32_t slot, int32_t item, wlx_ring* r, const int64_t* ranges, int32_t n_ranges,
                                   int32_t* n_frames_out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (!r || !ranges || n_ranges < 1) return fail(WLX_ERR_ARG, "wlx_logmel_ring: bad argument");
    if (n_ranges > WLX_LM_MAXRANGES) return fail(WLX_ERR_ARG, "wlx_logmel_ring: more than %d ranges", WLX_LM_MAXRANGES);
    if (item < 0 || item >= s->B) return fail(WLX_ERR_ARG, "bad item %d", item);
    if (r->device != e->device) return fail(WLX_ERR_ARG, "wlx_logmel_ring: the ring lives on device %d, the engine on %d", r->device, e->device);
    CK(hipSetDevice(e->device));
    std::lock_guard<std::mutex> lk(r->mu);
    long long tab[2 * WLX_LM_MAXRANGES];
    int64_t total = 0, prev_end = r->base;
    for (int i = 0; i < n_ranges; ++i) {
        const int64_t a = ranges[2 * i], b = ranges[2 * i + 1];
        if (a < prev_end || b <= a) return fail(a < r->base ? WLX_ERR_STATE : WLX_ERR_ARG, "wlx_logmel_ring: range %d = [%lld, %lld) is empty, out of order or no longer resident (ring starts at %lld)",
                                                i, (long long)a, (long long)b, (long long)r->base);
        if (b > r->base + r->resident) return fail(WLX_ERR_STATE, "wlx_logmel_ring: range %d ends at %lld, the ring at %lld", i, (long long)b, (long long)(r->base + r->resident));
        tab[2 * i] = a - r->base; tab[2 * i + 1] = total;
        total += b - a;
        prev_end = b;
    }
    if (total > 16000LL * 3600) return fail(WLX_ERR_ARG, "audio chunk too long");
    // a reader launched from ANOTHER stream may still be pending: its event is about to be re-recorded on this stream, so this stream first
    // waits for it (the new record then stands for both; one session = one slot = one stream makes this the rare case)
    if (r->read_pending && r->last_read_stream != s->stream) CK(hipStreamWaitEvent(s->stream, r->last_read, 0));
    CKR(flush_logmel(e, s));                         // requests recorded earlier go out first (one of them may be this item's)
    CKR(slot_grow_audio(e, s, (size_t)total));       // the feature buffers are sized with the audio buffers
    CK(hipMemcpyAsync(s->d_rng, tab, (size_t)n_ranges * 2 * sizeof(long long), hipMemcpyHostToDevice, s->stream));   // (pageable source: staged before the call returns)
    const int T = (int)((total + 160) / 160);
    LogmelBatch lb{};
    lb.n_items = 1; lb.pcm[0] = r->buf; lb.n[0] = (long)total; lb.feats[0] = s->feats + (size_t)item * e->spec.n_mels * s->feat_ld; lb.T[0] = T;
    lb.gmax[0] = s->gmax + item; lb.rng[0] = s->d_rng; lb.nr[0] = n_ranges;
    CK(hipEventRecord(s->ev_lm0, s->stream));
    launch_logmel_batch(lb, e->spec.n_mels, e->lm, s->feat_ld, s->stream);
    CK(hipGetLastError());
    CK(hipEventRecord(s->ev_lm1, s->stream));
    CK(hipEventRecord(r->last_read, s->stream));     // a later trim waits for this launch before it moves the samples
    r->read_pending = true;
    r->last_read_stream = s->stream;
    s->lm_pending = true;
    s->npcm[item] = 0;                               // the item's own PCM buffer does not hold this audio (wlx_logmel_resident would be wrong)
    s->nframes[item] = T;
    if (n_frames_out) *n_frames_out = T;
    return WLX_OK;
}

extern "C" int32_t wlx_logmel_resident(wlx_engine* e, int32_t slot, int32_t item, int32_t* n_frames_out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (item < 0 || item >= s->B) return fail(WLX_ERR_ARG, "bad item %d", item);
    const int64_t n = s->npcm[item];
    if (n <= 0) return fail(WLX_ERR_STATE, "item %d: no PCM resident (call wlx_pcm_put first)", item);
    CK(hipSetDevice(e->device));
    const int T = (int)((n + 160) / 160);
    float* dp = s->pcm + (size_t)item * s->pcm_cap;
    float* df = s->feats + (size_t)item * e->spec.n_mels * s->feat_ld;
    // No launch and no host wait here (round 4): the frame count is a function of the sample count; the request is recorded and goes out
    // together with the other items' (flush_logmel) in front of the first consumer. A batch of 12 windows used to be 36 launches and 12
    // launch -> wait round trips in front of its encoder; the launch's time is read lazily by wlx_timings_get.
    (void)dp; (void)df;
    if (std::find(s->lm_items.begin(), s->lm_items.end(), (int)item) == s->lm_items.end()) s->lm_items.push_back(item);
    s->nframes[item] = T;
    if (n_frames_out) *n_frames_out = T;
    return WLX_OK;
}

extern "C" int32_t wlx_logmel(wlx_engine* e, int32_t slot, int32_t item, const float* pcm, int64_t n,
                              int32_t* n_frames_out) {
    CKR(wlx_pcm_put(e, slot, item, pcm, n));
    return wlx_logmel_resident(e, slot, item, n_frames_out);
}

extern "C" int32_t wlx_features_get(wlx_engine* e, int32_t slot, int32_t item, float* out, int64_t cap_floats,
                                    int32_t* n_frames_out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (item < 0 || item >= s->B) return fail(WLX_ERR_ARG, "bad item");
    const int T = s->nframes[item], nm = e->spec.n_mels;
    if (n_frames_out) *n_frames_out = T;
    if (!out) return WLX_OK;
    CKR(flush_logmel(e, s));
    if ((int64_t)T * nm > cap_floats) return fail(WLX_ERR_ARG, "output buffer too small");
    CK(hipSetDevice(e->device));
    const float* df = s->feats + (size_t)item * nm * s->feat_ld;
    CK(hipMemcpy2DAsync(out, (size_t)T * 4, df, (size_t)s->feat_ld * 4, (size_t)T * 4, nm, hipMemcpyDeviceToHost, s->stream));
    CK(hipStreamSynchronize(s->stream));
    return WLX_OK;
}

extern "C" int32_t wlx_features_set(wlx_engine* e, int32_t slot, int32_t item, const float* feats,
                                    int32_t n_mels, int32_t n_frames) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (item < 0 || item >= s->B || !feats) return fail(WLX_ERR_ARG, "bad item / null");
    if (n_mels != e->spec.n_mels || n_frames < 1) return fail(WLX_ERR_ARG, "features must be [%d, T>=1]", e->spec.n_mels);
    CK(hipSetDevice(e->device));
    CKR(flush_logmel(e, s));                // (a recorded request for this item must not overwrite the features set here)
    CKR(slot_grow_audio(e, s, (size_t)n_frames * 160));
    float* df = s->feats + (size_t)item * n_mels * s->feat_ld;
    CK(hipMemcpy2DAsync(df, (size_t)s->feat_ld * 4, feats, (size_t)n_frames * 4, (size_t)n_frames * 4, n_mels,
                        hipMemcpyHostToDevice, s->stream));
    CK(hipStreamSynchronize(s->stream));
    s->nframes[item] = n_frames;
    return WLX_OK;
}

// ------------------------------------------------------------------------------------------------
// encoder
extern "C" int32_t wlx_encode(wlx_engine* e, int32_t slot, int32_t batch, const int32_t* seek, const int32_t* seg) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (batch < 1 || batch > s->B) return fail(WLX_ERR_ARG, "batch %d out of range (slot max %d)", batch, s->B);
    const wlx_spec& sp = e->spec;
    const int d = sp.d_model, F = sp.ffn, nm = sp.n_mels, H = e->H, T = WLX_T_AUDIO;
    CK(hipSetDevice(e->device));
    hipStream_t st = s->stream;
    // (every window validated BEFORE anything is recorded or launched: a refused call must leave the slot's timing events and
    // `en_pending` of the previous encode untouched — ADVICE r04)
    PrepWindows pw{};                       // the windows of the batch: one launch (blockIdx.y = item)
    for (int b = 0; b < batch; ++b) {
        const int sk = seek ? seek[b] : 0;
        int sg = seg ? seg[b] : (s->nframes[b] - sk);
        if (sk < 0 || sg < 0 || sk + sg > s->nframes[b])
            return fail(WLX_ERR_ARG, "item %d: window [%d,%d) outside %d feature frames", b, sk, sk + sg, s->nframes[b]);
        if (sg > WLX_N_FRAMES) sg = WLX_N_FRAMES;
        pw.seek[b] = sk; pw.seg[b] = sg;
    }
    CKR(flush_logmel(e, s));                // the recorded log-mel requests of the batch: one launch of each kernel
    CK(hipEventRecord(s->ev_en0, st));
    launch_prep_windows(s->feats, s->feat_ld, nm, (long)nm * s->feat_ld, pw, batch, s->featT, (long)s->featT_stride, st);
    GemmParams g{};
    // conv1 (k=3, s=1, p=1) + GELU: K-row of frame t = featT rows t..t+2 (row 0 / 3001 are the zero pad)
    g.A = s->featT; g.lda = nm; g.strideA = s->featT_stride; g.Wp = e->conv1_w; g.KT = e->conv1_KT;
    g.M = WLX_N_FRAMES; g.N = d; g.mode = GEMM_GELU_F16; g.bias = e->conv1_b;
    g.C = s->h1 + d; g.ldc = d; g.strideC = s->h1_stride;
    launch_gemm(g, batch, st);
    // conv2 (k=3, s=2, p=1) + GELU + sinusoidal positions -> fp32 residual stream
    g = GemmParams{};
    g.A = s->h1; g.lda = 2 * d; g.strideA = s->h1_stride; g.Wp = e->conv2_w; g.KT = 3 * d / 32;
    g.M = T; g.N = d; g.mode = GEMM_GELU_POS_F32; g.bias = e->conv2_b;
    g.X = s->x; g.ldx = d; g.strideX = (long)T * d; g.pos = e->enc_pos;
    launch_gemm(g, batch, st);
    const int M = batch * T;
    for (int l = 0; l < sp.enc_layers; ++l) {
        const EncLayerW& w = e->enc[l];
        launch_layernorm_f16(s->x, d, w.ln1_g, w.ln1_b, s->ln, d, M, d, st);
        g = GemmParams{};
        g.A = s->ln; g.lda = d; g.Wp = w.Wqkv; g.KT = d / 32; g.M = M; g.N = 3 * d; g.mode = GEMM_QKV; g.bias = w.bqkv;
        g.C = s->q; g.ldc = d; g.d = d; g.qscale = 0.125f; g.Kout = s->k; g.ldk = d; g.Vt = s->vt; g.ldvt = WLX_T_AUDIO_PAD;
        g.rows_per_item = T; g.kv_item_stride_k = (long)WLX_T_AUDIO_PAD * d; g.kv_item_stride_v = (long)d * WLX_T_AUDIO_PAD;
        launch_gemm(g, 1, st);
        launch_attn_encoder(s->q, d, s->k, d, s->vt, WLX_T_AUDIO_PAD, s->attn, d, T, H, batch, (long)T * d,
                            (long)WLX_T_AUDIO_PAD * d, (long)d * WLX_T_AUDIO_PAD, (long)T * d, st);
        g = GemmParams{};
        g.A = s->attn; g.lda = d; g.Wp = w.Wo; g.KT = d / 32; g.M = M; g.N = d; g.mode = GEMM_RESID_F32; g.bias = w.bo;
        g.X = s->x; g.ldx = d;
        launch_gemm(g, 1, st);
        launch_layernorm_f16(s->x, d, w.ln2_g, w.ln2_b, s->ln, d, M, d, st);
        g = GemmParams{};
        g.A = s->ln; g.lda = d; g.Wp = w.W1; g.KT = d / 32; g.M = M; g.N = F; g.mode = GEMM_GELU_F16; g.bias = w.b1;
        g.C = s->h2; g.ldc = F;
        launch_gemm(g, 1, st);
        g = GemmParams{};
        g.A = s->h2; g.lda = F; g.Wp = w.W2; g.KT = F / 32; g.M = M; g.N = d; g.mode = GEMM_RESID_F32; g.bias = w.b2;
        g.X = s->x; g.ldx = d;
        launch_gemm(g, 1, st);
    }
    launch_layernorm_f16_f32(s->x, d, e->enc_ln_g, e->enc_ln_b, s->enc16, s->enc32, d, M, d, st);
    // cross-attention K/V of every decoder layer in one GEMM (N = L*2d)
    g = GemmParams{};
    g.A = s->enc16; g.lda = d; g.Wp = e->Wckv; g.KT = d / 32; g.M = M; g.N = sp.dec_layers * 2 * d; g.mode = GEMM_CROSS_KV;
    g.bias = e->bckv; g.d = d; g.Kout = s->ck; g.ldk = d; g.Vt = s->cvt; g.ldvt = WLX_T_AUDIO_PAD; g.rows_per_item = T;
    g.kv_item_stride_k = (long)WLX_T_AUDIO_PAD * d; g.kv_item_stride_v = (long)d * WLX_T_AUDIO_PAD;
    g.kv_layer_stride_k = (long)s->B * WLX_T_AUDIO_PAD * d; g.kv_layer_stride_v = (long)s->B * d * WLX_T_AUDIO_PAD;
    launch_gemm(g, 1, st);
    CK(hipGetLastError());
    CK(hipEventRecord(s->ev_en1, st));
    // No host wait (round 4): everything that consumes the encoder output is ordered on the slot's stream, and the time is read lazily
    // by wlx_timings_get. The caller's next call — wlx_generate's host-side set-up, ~0.3 ms of staging and copies — now overlaps the
    // encoder on the GPU instead of following it.
    s->en_pending = true;
    s->enc_batch = batch;
    return WLX_OK;
}

extern "C" int32_t wlx_encoder_output_get(wlx_engine* e, int32_t slot, int32_t item, float* out, int64_t cap_floats) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (item < 0 || item >= s->enc_batch) return fail(WLX_ERR_STATE, "item %d not encoded", item);
    const size_t n = (size_t)WLX_T_AUDIO * e->spec.d_model;
    if (!out || (int64_t)n > cap_floats) return fail(WLX_ERR_ARG, "output buffer too small");
    CK(hipSetDevice(e->device));
    CK(hipMemcpyAsync(out, s->enc32 + (size_t)item * n, n * 4, hipMemcpyDeviceToHost, s->stream));
    CK(hipStreamSynchronize(s->stream));
    return WLX_OK;
}

// ------------------------------------------------------------------------------------------------
// per-kernel profiler hook (wlx_debug_profile_step): when s->prof is set a launch of the decoder pass is either only
// listed (name as rocprofv3 prints it + its ALGORITHMIC bytes: weights / K,V it has to stream once) or filtered by name.
template <class F>
static inline void plaunch(Slot* s, const char* name, double bytes, F&& f) {
    if (!s->prof) { f(); return; }
    if (s->prof->list_only) { s->prof->recs.push_back(ProfRec{name, bytes}); return; }
    if (s->prof->only == name) f();
}
static double gemv_bytes(const GemvParams& p) { return 2.0 * (double)p.N * (double)p.K + (p.bias ? 4.0 * p.N : 0.0); }
// three or more live slots on the slot's device: the GPU is work-bound there and the decode projections pick work-saving
// launch shapes (GemvParams::busy_device). Evaluated when a step graph is captured / looked up: the variant is part of the graph key.
static bool device_is_busy(const Slot* s) {
    return s->device_of >= 0 && s->device_of < 64 && g_slots_live[s->device_of].load(std::memory_order_relaxed) >= 3;
}
static void pgemv(Slot* s, const GemvParams& p0) {
    GemvParams p = p0;
    p.busy_device = s->busy_variant ? 1 : 0;
    if (!s->prof) { launch_dec_gemv(p, s->stream); return; }
    const std::string nm = dec_gemv_kernel_name(p);
    plaunch(s, nm.c_str(), gemv_bytes(p), [&] { launch_dec_gemv(p, s->stream); });
}

// ------------------------------------------------------------------------------------------------
// decoder pass: embed -> L x {self-attn block, cross-attn block, MLP} -> (final LN + vocab projection)
// Row tables / ancestry must already be on the device. `rows` live rows in `groups` groups of R rows.
static void decoder_pass(Engine* e, Slot* s_, int rows, int R, int groups, bool with_logits, bool check_done, const Slot::DecBufs* alt = nullptr) {
    const wlx_spec& sp = e->spec;
    const int d = sp.d_model, F = sp.ffn, H = e->H;
    // the working set: the slot's own (decode steps, chunked passes) or the prompt-prefill set (alt); everything else
    // (stream, KV caches, cross K/V, ancestry, profiler hook) is the slot's
    struct View {
        Slot* base; float* xd; half_t *qd, *attnd, *hd; float* slab; long slab_rows; half_t* part_o; float* part_ml;
        int *d_token, *d_pos, *d_cache, *d_ancrow, *d_group_item;
        Slot* operator->() const { return base; }
    } s{s_, alt ? alt->xd : s_->xd, alt ? alt->qd : s_->qd, alt ? alt->attnd : s_->attnd, alt ? alt->hd : s_->hd,
        alt ? alt->slab : s_->slab, alt ? (long)alt->slab_rows : (long)s_->rows_cap, alt ? alt->part_o : s_->part_o, alt ? alt->part_ml : s_->part_ml,
        alt ? alt->d_token : s_->d_token, alt ? alt->d_pos : s_->d_pos, alt ? alt->d_cache : s_->d_cache,
        alt ? alt->d_ancrow : s_->d_ancrow, alt ? alt->d_group_item : s_->d_group_item};
    hipStream_t st = s->stream;
    // The decoder pass only rewrites scratch and re-appends the same K/V at the same position when it runs once
    // more after the search has raised `done` (the host runs at most one step ahead), so the second-generation
    // kernels do not test the flag: the test was a dependent scalar load at the head of ~100 launches per step.
    const int* done = (check_done && g_decode_v1) ? s->st.done : nullptr;
    RowTables rt{s.d_token, s.d_pos, s.d_cache, s.d_ancrow, s->d_anc, s->d_intok};
    const long crs = (long)WLX_T_TEXT * d;
    // ---- the parameter sets of one layer (the first projection's residual source is filled in below)
    auto qkv_params = [&](int l, int xsrc) {
        const DecLayerW& w = e->dec[l];
        GemvParams p{};
        p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_QKV; p.M = rows; p.K = d; p.KT = d / 32; p.N = 3 * d;
        p.Wp = w.Wqkv; p.bias = w.bqkv; p.X = s.xd; p.ldx = d; p.gamma = w.ln1_g; p.beta = w.ln1_b;
        p.Yh = s.qd; p.ldyh = d; p.d = d; p.qscale = 0.125f;
        p.Kc = s->kc + (size_t)l * s->cache_rows * crs; p.Vc = s->vc + (size_t)l * s->cache_rows * crs; p.cache_row_stride = crs;
        p.row_cache = s.d_cache; p.row_pos = s.d_pos; p.done = done;
        p.xsrc = xsrc; p.slab = s.slab; p.slab_stride = s.slab_rows * d;
        if (xsrc == GEMV_X_EMBED) {
            p.tok_emb = e->tok_emb16; p.pos_emb = e->dec_pos; p.emb_token = s.d_token; p.intok = s->d_intok;
            p.Xres = s.xd; p.ldxres = d;
        }
        return p;
    };
    auto oproj_params = [&](int l, int xsrc) {
        const DecLayerW& w = e->dec[l];
        GemvParams p{};
        p.in_mode = GEMV_IN_F16; p.out_mode = GEMV_OUT_RESID; p.M = rows; p.K = d; p.KT = d / 32; p.N = d;
        p.Wp = w.Wo; p.bias = w.bo; p.Xh = s.attnd; p.ldxh = d; p.Xres = s.xd; p.ldxres = d; p.qscale = 1.f; p.done = done;
        p.xsrc = xsrc; p.slab = s.slab; p.slab_stride = s.slab_rows * d;
        return p;
    };
    auto vocab_params = [&](int xsrc) {
        GemvParams p{};
        p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_F32; p.M = rows; p.K = d; p.KT = d / 32; p.N = sp.vocab;
        p.Wp = e->Wvocab; p.bias = nullptr; p.X = s.xd; p.ldx = d; p.gamma = e->dec_ln_g; p.beta = e->dec_ln_b;
        p.Y = s->logits; p.ldy = s->ldl; p.qscale = 1.f; p.done = done;
        p.xsrc = xsrc; p.slab = s.slab; p.slab_stride = s.slab_rows * d;
        return p;
    };
    // ---- what this pass may use (decided once, from the shapes only — never from what a profiling filter lets through):
    // the K-split MLP output projection needs every consumer of its slabs on the lean kernel; the embedding folds into
    // layer 0's first projection under the same condition (re-measured in round 5 against the separate embedding launch: 27.80 vs 28.02 ms
    // per window, profiles/r5a_*).
    int KS = dec_gemv_slab_split(rows, F, d);
    if (rows > s.slab_rows) KS = 0;         // the partial-sum slabs of this working set hold slab_rows rows
    // batched decode steps (17..64 rows) keep the single MLP output launch (round 4, profiles/r4b_*): the split saves ~1 us
    // there but every consumer of the slabs then reads three fp32 copies of every row in its LayerNorm prologue — per 16-column
    // workgroup — and those launches are bound by load instructions per CU (60 rows, small.en: first projection 9.8 us with
    // slabs, 7.0 us without; large-v3 at 40 rows: 11.4 -> 8.7 us).
    if (rows > 16 && !alt) KS = 0;
    // ... and the one-pass prompt prefill (round 6): its K-split MLP projection made every 16-column workgroup of the next layer's first projection
    // (144 x 14 workgroups at 224 rows) read THREE fp32 copies of its rows and lose the four-column-tile form: 21.8 us per launch against 11 us.
    // 224 tokens: 1.099 -> 0.955 ms (Whisper-small), 6.66 -> 5.31 ms (large-v3), profiles/r6g_prefill_time.txt. (The joint pass over a batch's
    // short prompts keeps its form: its rows must stay bit-identical to the single calls' chunked prefill.)
    if (alt && s->pf_one_pass) KS = 0;
    if (KS && !(dec_gemv_is_lean(qkv_params(0, GEMV_X_SLABS)) && dec_gemv_is_lean(oproj_params(0, GEMV_X_SLABS)))) KS = 0;
    // (batched steps, 17..64 rows: the folded form gathers ONE row per wave and trip — two trips per 16-row tile, the second behind the
    // weight stream — and has no four-tile instantiation: 13.6 us at 60 rows against 2.4 + 5.9 us for the embedding launch + the plain
    // four-tile projection, profiles/r4s_decode_step.txt.)
    // (prefill passes of more than 64 rows keep the embedding launch, as they always did: the folded form's K split over the waves — one
    // row per wave — differs from the plain LayerNorm projection's, i.e. another summation order for the prompt rows' layer-0 K / V)
    const bool fold_embed = (rows <= 16 || (alt != nullptr && rows <= 64)) && dec_gemv_is_lean(qkv_params(0, GEMV_X_EMBED));
    // (Measured and closed: ONE LayerNorm launch per layer phase + fp16-rows-in projections instead of the LayerNorm prologue in every
    // 16-column workgroup — for batched decode steps in round 4, for the prompt-prefill pass in round 5 (conditioned window 30.88 vs
    // 30.89 ms, profiles/r5a_*): no gain either way; the kernel left the library.)
    if (!fold_embed)
        plaunch(s.base, "dec_embed_kernel", (double)rows * d * (2 + 4), [&] { launch_dec_embed(e->tok_emb16, e->dec_pos, d, rt, rows, s.xd, done, st); });
    bool slabs_pending = false;             // the residual stream is xd + slabs until the next residual update writes the sum back
    for (int l = 0; l < sp.dec_layers; ++l) {
        const DecLayerW& w = e->dec[l];
        half_t* kc = s->kc + (size_t)l * s->cache_rows * crs;
        half_t* vc = s->vc + (size_t)l * s->cache_rows * crs;
        // LN1 + QKV, K/V appended to the self-attention cache
        {
            GemvParams pq = qkv_params(l, (l == 0 && fold_embed) ? GEMV_X_EMBED : (slabs_pending ? GEMV_X_SLABS : GEMV_X_PLAIN));
            pgemv(s.base, pq);
        }
        plaunch(s.base, "dec_self_attn2_kernel", 4.0 * rows * d * (s->prof ? s->prof->t + 1 : 1), [&] { launch_dec_self_attn(s.qd, d, kc, vc, crs, d, H, rt, rows, s.attnd, d, done, s->anc_ident, st); });
        pgemv(s.base, oproj_params(l, slabs_pending ? GEMV_X_SLABS : GEMV_X_PLAIN));
        slabs_pending = false;
        GemvParams p{};
        // LN2 + cross-attention query + cross-attention partials: one fused launch when the shape allows and nobody needs
        // the query rows (word alignment captures them), else projection and attention separately
        const half_t* ckl = s->ck + (size_t)l * s->B * WLX_T_AUDIO_PAD * d;
        const half_t* cvl = s->cvt + (size_t)l * s->B * d * WLX_T_AUDIO_PAD;
        // (not for the one-pass prompt prefill: with 14+ groups every (split, head, group) workgroup re-reads its head's 96 KiB of
        // query weights — 224 tokens: 1.49 ms fused, 1.32 ms as projection + attention, profiles/r3l_prefill_fused_cq.txt)
        const bool fused = !s->align && rows <= 48 && dec_cq_cross_attn_eligible(d, H, R);
        if (fused)
            plaunch(s.base, "dec_cq_cross_attn_kernel", 2.0 * d * d + 4.0 * groups * WLX_T_AUDIO * d, [&] {
                launch_dec_cq_cross_attn(s.xd, d, w.ln2_g, w.ln2_b, w.Wcq, w.bcq, 0.125f, d, ckl, cvl, (long)WLX_T_AUDIO_PAD * d, H, R,
                                         groups, rows, s.d_group_item, s.part_o, s.part_ml, st);
            });
        if (!fused) {
            p = GemvParams{};
            p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_F16; p.M = rows; p.K = d; p.KT = d / 32; p.N = d;
            p.Wp = w.Wcq; p.bias = w.bcq; p.X = s.xd; p.ldx = d; p.gamma = w.ln2_g; p.beta = w.ln2_b;
            p.Yh = s.qd; p.ldyh = d; p.qscale = 0.125f; p.done = done;
            pgemv(s.base, p);
            if (s->align) {     // word alignment: raw q.k of this layer's alignment heads for the rows of this chunk
                const Slot::AlignCapture& a = *s->align;
                for (int hi = 0; hi < a.n_heads; ++hi)
                    if (a.heads[2 * hi] == l)
                        launch_dec_align_scores(s.qd, d, s->ck + ((size_t)l * s->B + a.item) * WLX_T_AUDIO_PAD * d, a.heads[2 * hi + 1], rows,
                                                a.scores + ((size_t)hi * a.n_tok + a.row0) * WLX_T_AUDIO_PAD, st);
            }
            plaunch(s.base, "dec_cross_attn_kernel", 4.0 * groups * WLX_T_AUDIO * d, [&] {
                launch_dec_cross_attn(s.qd, d, ckl, cvl, (long)WLX_T_AUDIO_PAD * d, H, R, groups, rows, s.d_group_item, s.part_o, s.part_ml, st);
            });
        }
        p = GemvParams{};
        p.in_mode = GEMV_IN_XATTN; p.out_mode = GEMV_OUT_RESID; p.M = rows; p.K = d; p.KT = d / 32; p.N = d;
        p.Wp = w.Wco; p.bias = w.bco; p.part_o = s.part_o; p.part_ml = s.part_ml; p.H = H; p.R = R;
        p.Xres = s.xd; p.ldxres = d; p.qscale = 1.f; p.done = done;
        if (rows > 16) {
            // batched rows: the split combine once, in its own launch, then a plain fp16-rows-in projection (decoder.hip)
            plaunch(s.base, "dec_xattn_combine_kernel", 0.0, [&] { launch_dec_xattn_combine(s.part_o, s.part_ml, rows, H, R, s.attnd, d, st); });
            p.in_mode = GEMV_IN_F16; p.Xh = s.attnd; p.ldxh = d;
        }
        pgemv(s.base, p);
        // LN3 + MLP
        p = GemvParams{};
        p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_GELU_F16; p.M = rows; p.K = d; p.KT = d / 32; p.N = F;
        p.Wp = w.W1; p.bias = w.b1; p.X = s.xd; p.ldx = d; p.gamma = w.ln3_g; p.beta = w.ln3_b;
        p.Yh = s.hd; p.ldyh = F; p.qscale = 1.f; p.done = done;
        pgemv(s.base, p);
        p = GemvParams{};
        p.in_mode = GEMV_IN_F16; p.out_mode = GEMV_OUT_RESID; p.M = rows; p.K = F; p.KT = F / 32; p.N = d;
        p.Wp = w.W2; p.bias = w.b2; p.Xh = s.hd; p.ldxh = F; p.Xres = s.xd; p.ldxres = d; p.qscale = 1.f; p.done = done;
        // the last layer: its consumer is the vocabulary projection. While that was 1621 workgroups that would each have summed the slabs the
        // split was a loss there (+2.2 us against 1.2 us saved, profiles/r2f_*) and the last layer kept the single launch — 11-13 us at 5 rows,
        // the slowest projection of the step (profiles/r5b_*). dec_vocab_kernel (203 workgroups, one LayerNorm each) takes rows + slabs.
        const bool last_split = with_logits && dec_gemv_is_lean(vocab_params(GEMV_X_SLABS));
        if (KS && (l + 1 < sp.dec_layers || last_split)) {
            p.out_mode = GEMV_OUT_SLAB; p.KTS = p.KT / KS; p.slab = s.slab; p.slab_stride = s.slab_rows * d;
            slabs_pending = true;
        }
        pgemv(s.base, p);
    }
    if (with_logits) pgemv(s.base, vocab_params(slabs_pending ? GEMV_X_SLABS : GEMV_X_PLAIN));
}

// upload row tables for a pass: token/pos/cache/ancrow [rows], group_item [groups]
static int upload_rows(Slot* s, const std::vector<int>& token, const std::vector<int>& pos,
                       const std::vector<int>& cache, const std::vector<int>& ancrow, const std::vector<int>& group_item,
                       int* own_staging = nullptr) {
    const size_t rows = token.size(), ng = group_item.size();
    if (4 * rows + ng > s->h_stage_ints) return fail(WLX_ERR_ARG, "row table too large");
    s->anc_ident = true;                                    // every row reads its history through its own ancestry row
    for (size_t i = 0; i < rows; ++i) s->anc_ident = s->anc_ident && ancrow[i] == (int)i;
    // the shared staging buffer may still be the source of an earlier pass's copies: wait; a caller that brings its own
    // pinned area (wlx_generate: written once per call) does not have to
    if (!own_staging) CK(hipStreamSynchronize(s->stream));
    int* h = own_staging ? own_staging : s->h_stage;
    memcpy(h, token.data(), rows * 4); memcpy(h + rows, pos.data(), rows * 4);
    memcpy(h + 2 * rows, cache.data(), rows * 4); memcpy(h + 3 * rows, ancrow.data(), rows * 4);
    memcpy(h + 4 * rows, group_item.data(), ng * 4);
    CK(hipMemcpyAsync(s->d_token, h, rows * 4, hipMemcpyHostToDevice, s->stream));
    CK(hipMemcpyAsync(s->d_pos, h + rows, rows * 4, hipMemcpyHostToDevice, s->stream));
    CK(hipMemcpyAsync(s->d_cache, h + 2 * rows, rows * 4, hipMemcpyHostToDevice, s->stream));
    CK(hipMemcpyAsync(s->d_ancrow, h + 3 * rows, rows * 4, hipMemcpyHostToDevice, s->stream));
    CK(hipMemcpyAsync(s->d_group_item, h + 4 * rows, ng * 4, hipMemcpyHostToDevice, s->stream));
    return WLX_OK;
}

// prefill `n` tokens of one sequence (audio item `item`, KV-cache row `crow`) starting at position pos0,
// in chunks of <= 48 rows (below); if logits_out != null the vocabulary projection runs and rows are copied out.
static int prefill_tokens(Engine* e, Slot* s, int item, int crow, const int* tokens, int pos0, int n,
                          float* logits_host, int nsp_index, int nsp_token, float* nsp_out) {
    // ONE pass over the whole prompt (round 3): every window after the first carries up to 224 prompt tokens
    // (transcriber_faster_whisper.py:1480-1513) and the chunked form below costs a full ~87-launch decoder pass per 48 rows —
    // 4.6 ms for 224 tokens on Whisper-small, 18 ms on large-v3 (profiles/r3f_prefill_time.txt), 16-19 % of a window. Here
    // the pass runs ONCE over all rows: each projection is one launch of the lean kernel whose grid.z walks 48-row chunks
    // (the weight tiles are re-read from L2 by the chunks), attention and combine launches take the rows as they are.
    // Logits are only needed at the <|startoftranscript|> row (no_speech_prob); callers that want every row's logits
    // (debug hook) keep the chunked form.
    const bool one_pass = [] { const char* v = getenv("WLX_PREFILL_ONE_PASS"); return !(v && v[0] == '0'); }();   // (read per call: the A/B test toggles it)
    if (one_pass && s->pf_ok && !logits_host && !s->align && !s->prof && n > 48 && n <= WLX_T_TEXT && !g_decode_v1) {
        const int rows = n, groups = (rows + 15) / 16;
        // its own pinned staging (round 6): the shared one needed a wait for the stream first — "may still feed an earlier pass's copies" — which
        // let the GPU idle between the encoder's end and this pass while the host built and uploaded the tables. Only a SECOND one-pass prefill of
        // the same call (a batch of long prompts) has to wait: wlx_generate clears the flag, and a call returns after its own uploads were consumed.
        if (s->h_pf_used) CK(hipStreamSynchronize(s->stream));
        s->h_pf_used = true;
        int* h = s->h_pf;
        for (int i = 0; i < rows; ++i) { h[i] = tokens[i]; h[rows + i] = pos0 + i; h[2 * rows + i] = crow; h[3 * rows + i] = crow; }
        for (int g = 0; g < groups; ++g) h[4 * rows + g] = item;
        const Slot::DecBufs& b = s->pf;
        CK(hipMemcpyAsync(b.d_token, h, rows * 4, hipMemcpyHostToDevice, s->stream));
        CK(hipMemcpyAsync(b.d_pos, h + rows, rows * 4, hipMemcpyHostToDevice, s->stream));
        CK(hipMemcpyAsync(b.d_cache, h + 2 * rows, rows * 4, hipMemcpyHostToDevice, s->stream));
        CK(hipMemcpyAsync(b.d_ancrow, h + 3 * rows, rows * 4, hipMemcpyHostToDevice, s->stream));
        CK(hipMemcpyAsync(b.d_group_item, h + 4 * rows, groups * 4, hipMemcpyHostToDevice, s->stream));
        s->anc_ident = false;                                       // every row reads its history through the prompt's cache row
        s->pf_one_pass = true;
        decoder_pass(e, s, rows, 16, groups, false, false, &b);
        s->pf_one_pass = false;
        CK(hipGetLastError());
        if (nsp_index >= 0 && nsp_index < rows) {
            const int d = e->spec.d_model;
            GemvParams p{};
            p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_F32; p.M = 1; p.K = d; p.KT = d / 32; p.N = e->spec.vocab;
            p.Wp = e->Wvocab; p.bias = nullptr; p.X = b.xd + (size_t)nsp_index * d; p.ldx = d; p.gamma = e->dec_ln_g; p.beta = e->dec_ln_b;
            p.Y = s->logits; p.ldy = s->ldl; p.qscale = 1.f; p.xsrc = GEMV_X_PLAIN;
            launch_dec_gemv(p, s->stream);
            launch_token_prob(s->logits, s->ldl, e->spec.vocab, 1, nsp_token, s->d_tokprob, s->stream);
            CK(hipMemcpyAsync(nsp_out, s->d_tokprob, 4, hipMemcpyDefault, s->stream));       // (nsp_out: the pinned result area of wlx_generate, or device memory)
            CK(hipGetLastError());
        }
        return WLX_OK;
    }
    // chunk size: 48 rows = three 16-row MFMA tiles = one launch of the lean projections per chunk (a 64-row chunk runs them as
    // two row chunks in grid.z); WLX_PREFILL_ROWS=16..320 (A/B, tests)
    // (read per call: tests toggle it; up to the slot's row capacity — 120 teacher-forced rows in one pass on a 24 x 5 slot)
    const int chunk = std::min(s->rows_cap, [] { const char* v = getenv("WLX_PREFILL_ROWS"); const int c = v ? atoi(v) : 48; return (c >= 16 && c <= WLX_MAX_DEC_ROWS) ? c : 48; }());
    for (int c0 = 0; c0 < n; c0 += chunk) {
        const int rows = std::min(chunk, n - c0);
        const int groups = (rows + 15) / 16;
        std::vector<int> tk(rows), ps(rows), ca(rows, crow), an(rows, crow), gi(groups, item);
        for (int i = 0; i < rows; ++i) { tk[i] = tokens[c0 + i]; ps[i] = pos0 + c0 + i; }
        CKR(upload_rows(s, tk, ps, ca, an, gi));
        const bool want_nsp = nsp_index >= c0 && nsp_index < c0 + rows;
        const bool lg = (logits_host != nullptr) || want_nsp;
        decoder_pass(e, s, rows, 16, groups, lg, false);
        CK(hipGetLastError());
        if (logits_host) {
            CK(hipMemcpy2DAsync(logits_host + (size_t)c0 * e->spec.vocab, (size_t)e->spec.vocab * 4, s->logits,
                                (size_t)s->ldl * 4, (size_t)e->spec.vocab * 4, rows, hipMemcpyDeviceToHost, s->stream));
        }
        if (want_nsp) {
            launch_token_prob(s->logits + (size_t)(nsp_index - c0) * s->ldl, s->ldl, e->spec.vocab, 1, nsp_token,
                              s->d_tokprob, s->stream);
            CK(hipMemcpyAsync(nsp_out, s->d_tokprob, 4, hipMemcpyDeviceToDevice, s->stream));
        }
    }
    return WLX_OK;
}

// identity ancestry for cache row `crow` over positions [0, upto)
static int set_anc_rows(Slot* s, const std::vector<short>& anc_host, int first_row, int nrows) {
    CK(hipMemcpyAsync(s->d_anc + (size_t)first_row * WLX_T_TEXT, anc_host.data(), (size_t)nrows * WLX_T_TEXT * sizeof(short),
                      hipMemcpyHostToDevice, s->stream));
    CK(hipStreamSynchronize(s->stream));   // anc_host is caller stack memory
    return WLX_OK;
}

// token search of one step: beam mode runs the chunked scan + merge/update pair, sampling (T > 0 fallback)
// the one-workgroup-per-row kernels
static void launch_search(Engine* e, Slot* s, int rows, int R, int groups, bool sampling) {
    if (sampling || g_decode_v1) {
        launch_search_rows(s->logits, s->d_sp, rows, s->st, s->stream);
        launch_search_update(s->d_sp, groups, s->st, s->stream);
    } else {
        launch_search_scan3(s->logits, s->ldl, e->spec.vocab, s->d_sp, rows, s->st, s->stream);
        launch_search_merge_update3(s->logits, s->ldl, e->spec.vocab, s->d_sp, groups, R, s->st, s->stream);
    }
}

// Steps per graph launch (round 6). The boundary between two step graphs is 8.6-8.9 us of idle GPU against 1.45 us between two kernels of one graph
// (profiles/r6ab_chunk_timeline_under_rocprof.txt): with TWO decode steps per graph every second boundary is an ordinary kernel boundary, -3.6 us per
// step. The price is how far a decode can run past its end: the host enqueues graph g + 1 when the LAST step of graph g has started (the update kernels
// mirror their step number to pinned memory), so an end of text in the first step of a graph leaves one scratch-only step behind it and one in the second
// step leaves two — 1.5 on average against 1 with one step per graph. The call itself returns at the done word either way (the results are in pinned
// memory); only work queued behind it on the same stream sees the extra 0.2 ms. So two steps are used where latency is what counts and the GPU has
// room — ONE live slot on the device, one stream's rows (<= 16) — and one step everywhere else (several clients' slots, batched rows).
// WLX_GRAPH_STEPS=1 forces one step per graph (A/B).
static int graph_steps_for(const Slot* s, int rows) {
    static const int env = [] { const char* v = wlx_ab("WLX_GRAPH_STEPS"); const int n = v ? atoi(v) : 2; return (n == 1 || n == 2) ? n : 2; }();
    if (env == 1 || rows > 16) return 1;
    const bool alone = s->device_of >= 0 && s->device_of < 64 && g_slots_live[s->device_of].load(std::memory_order_relaxed) <= 1;
    return alone ? 2 : 1;
}

static int get_step_graph(Engine* e, Slot* s, int rows, int R, int groups, bool sampling, int nsteps, hipGraphExec_t* out) {
    s->busy_variant = device_is_busy(s);
    StepGraphKey key{rows, R, groups * 8 + (nsteps == 2 ? 4 : 0) + (s->busy_variant ? 2 : 0) + (sampling ? 1 : 0)};
    auto it = s->graphs.find(key);
    if (it != s->graphs.end()) { *out = it->second; return WLX_OK; }
    hipGraph_t graph;
    // One eager pass first: the first launch of a kernel instantiation may have to raise its dynamic-LDS limit
    // (decoder.hip g2_launch), which must not happen inside a capture. It only rewrites scratch and re-appends the K/V the
    // captured replay appends again (same rows, same positions); the search, which mutates state, is not run.
    decoder_pass(e, s, rows, R, groups, true, true);
    CK(hipGetLastError());
    CK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < nsteps; ++k) {
        decoder_pass(e, s, rows, R, groups, true, true);
        launch_search(e, s, rows, R, groups, sampling);
    }
    const hipError_t ce = hipStreamEndCapture(s->stream, &graph);
    if (ce != hipSuccess) {
        // An invalidated capture leaves the stream refusing every later operation ("previous error during capture"), so
        // one bad capture would end the client's session for good. Replace the stream: this call fails (the session logs
        // it and moves on to the next chunk, whisper_live/backend/base.py:134-137), the next one captures afresh.
        (void)hipGetLastError();
        hipStream_t ns = nullptr;
        bool ded = false;
        if (s->dedicated_queue) { g_dedicated_live[e->device].fetch_sub(1); s->dedicated_queue = false; }
        if (create_slot_stream(e->device, &ns, &ded) == WLX_OK) {
            s->dedicated_queue = ded;
            (void)hipStreamDestroy(s->stream);
            s->stream = ns;
        }
        return fail(WLX_ERR_HIP, "decode-step graph capture failed: %s (slot stream replaced)", hipGetErrorString(ce));
    }
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphDestroy(graph));
    s->graphs[key] = exec;
    *out = exec;
    return WLX_OK;
}

// `nsteps` decode steps (1 or 2): one graph launch, or the same launches eagerly
static int run_step(Engine* e, Slot* s, int rows, int R, int groups, bool sampling, int nsteps = 1) {
    if (e->use_graph) {
        hipGraphExec_t exec;
        CKR(get_step_graph(e, s, rows, R, groups, sampling, nsteps, &exec));
        CK(hipGraphLaunch(exec, s->stream));
    } else {
        s->busy_variant = device_is_busy(s);
        for (int k = 0; k < nsteps; ++k) {
            decoder_pass(e, s, rows, R, groups, true, true);
            launch_search(e, s, rows, R, groups, sampling);
        }
        CK(hipGetLastError());
    }
    return WLX_OK;
}

static int fill_search_params(Engine* e, Slot* s, int batch, int R, const wlx_gen_opts* o, bool apply_ts, SearchParams* sp) {
    memset(sp, 0, sizeof(*sp));
    sp->V = e->spec.vocab; sp->ldl = s->ldl; sp->items = batch; sp->R = R; sp->rows = batch * R;
    sp->sampling = (o->sampling_temperature > 0.f || o->beam_size <= 1) ? 1 : 0;
    sp->beam = sp->sampling ? 1 : o->beam_size;
    sp->ncand = sp->sampling ? 1 : 2 * o->beam_size;
    sp->max_cand_hyp = std::max(1, (int)std::lround((double)o->beam_size * (double)o->patience));
    sp->num_hyp = std::max(1, o->num_hypotheses);
    sp->allow_early_exit = (o->length_penalty == 0.f) ? 1 : 0;
    sp->length_penalty = o->length_penalty; sp->rep_penalty = (o->repetition_penalty > 0.f) ? o->repetition_penalty : 1.f;
    sp->temperature = o->sampling_temperature; sp->no_repeat_ngram = o->no_repeat_ngram_size;
    sp->topk = (o->sampling_temperature > 0.f) ? o->sampling_topk : 1;
    sp->suppress_blank = o->suppress_blank; sp->apply_ts_rules = apply_ts ? 1 : 0;
    sp->max_initial_ts = o->max_initial_timestamp_index;
    sp->sot = o->ids.sot; sp->eot = o->ids.eot; sp->no_timestamps = o->ids.no_timestamps; sp->ts_begin = o->ids.timestamp_begin;
    sp->no_speech = o->ids.no_speech; sp->blank = o->ids.blank;
    sp->max_length = o->max_length; sp->seed = o->seed; sp->suppress_mask = s->d_suppress;
    return WLX_OK;
}

static int validate_opts(Engine* e, Slot* s, int batch, const wlx_gen_opts* o) {
    if (!o) return fail(WLX_ERR_ARG, "null opts");
    const int V = e->spec.vocab;
    auto bad = [&](int id) { return id < 0 || id >= V; };
    if (bad(o->ids.sot) || bad(o->ids.eot) || bad(o->ids.no_timestamps) || bad(o->ids.timestamp_begin) || bad(o->ids.no_speech))
        return fail(WLX_ERR_ARG, "token ids out of vocabulary");
    if (o->max_length < 2 || o->max_length > WLX_T_TEXT) return fail(WLX_ERR_ARG, "max_length must be 2..448");
    const bool sampling = (o->sampling_temperature > 0.f || o->beam_size <= 1);
    const int R = sampling ? std::max(1, o->num_hypotheses) : o->beam_size;
    if (R < 1 || R > s->R) return fail(WLX_ERR_ARG, "beam_size/num_hypotheses %d exceeds slot rows per item %d", R, s->R);
    if (!sampling && o->num_hypotheses > o->beam_size) return fail(WLX_ERR_ARG, "num_hypotheses > beam_size");
    if (!sampling && 2 * o->beam_size > WLX_MAX_CAND) return fail(WLX_ERR_ARG, "beam_size too large");
    if (batch * R > s->rows_cap) return fail(WLX_ERR_ARG, "too many decoder rows");
    if (o->n_suppress_tokens < 0 || (o->n_suppress_tokens > 0 && !o->suppress_tokens)) return fail(WLX_ERR_ARG, "bad suppress_tokens");
    return WLX_OK;
}

static int upload_suppress(Engine* e, Slot* s, const wlx_gen_opts* o) {
    const int V = e->spec.vocab, words = (V + 31) / 32;
    // a session passes the same list on every call: the mask on the device is then already the right one
    if (s->suppress_valid && (int)s->last_suppress.size() == o->n_suppress_tokens &&
        (o->n_suppress_tokens == 0 || memcmp(s->last_suppress.data(), o->suppress_tokens, (size_t)o->n_suppress_tokens * 4) == 0))
        return WLX_OK;
    s->suppress_valid = false;
    std::vector<unsigned> mask(words, 0u);
    for (int i = 0; i < o->n_suppress_tokens; ++i) {
        const int id = o->suppress_tokens[i];
        if (id >= 0 && id < V) mask[id >> 5] |= 1u << (id & 31);
    }
    CK(hipMemcpyAsync(s->d_suppress, mask.data(), words * 4, hipMemcpyHostToDevice, s->stream));
    CK(hipStreamSynchronize(s->stream));
    s->last_suppress.assign(o->suppress_tokens, o->suppress_tokens + o->n_suppress_tokens);
    s->suppress_valid = true;
    return WLX_OK;
}

struct HypOut { std::vector<int> tokens; float score; };

static int generate_impl(Engine* e, Slot* s, int batch, const int32_t* prompts, const int32_t* plens, int pstride,
                         const int32_t* enc_items, const wlx_gen_opts* o, bool injected_logits, const float* inj, int inj_steps,
                         int32_t* tokens_out, int tstride, int32_t* n_tokens_out, float* scores_out, float* nsp_out) {
    CKR(validate_opts(e, s, batch, o));
    static const bool gen_trace = getenv("WLX_GEN_TRACE") != nullptr;
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tg0 = now_us();
    double tg_launch = 0.0, tg_wait = 0.0;
    s->h_pf_used = false;
    const bool sampling = (o->sampling_temperature > 0.f || o->beam_size <= 1);
    const int R = sampling ? std::max(1, o->num_hypotheses) : o->beam_size;
    const int rows = batch * R, V = e->spec.vocab;
    hipStream_t st = s->stream;
    int max_steps = 0;
    bool apply_ts = true;
    for (int b = 0; b < batch; ++b) {
        const int pl = plens[b];
        if (pl < 1 || pl >= o->max_length) return fail(WLX_ERR_ARG, "item %d: prompt length %d vs max_length %d", b, pl, o->max_length);
        const int32_t* pr = prompts + (size_t)b * pstride;
        for (int i = 0; i < pl; ++i) {
            if (pr[i] < 0 || pr[i] >= V) return fail(WLX_ERR_ARG, "prompt token out of vocabulary");
            if (pr[i] == o->ids.no_timestamps) apply_ts = false;   // CT2: timestamp rules only without <|notimestamps|>
        }
        max_steps = std::max(max_steps, o->max_length - pl);
    }
    CK(hipEventRecord(s->ev0, st));
    CKR(upload_suppress(e, s, o));
    // ---- per-call state: everything the host prepares goes through the slot's pinned staging (h_gen) — asynchronous
    // copies with no synchronisation in between (pageable sources made every one of them a blocking, bounce-buffered copy)
    // — and the zeroing is one launch. The staging is only rewritten by the next call, after this one's final wait.
    SearchState& S = s->st;
    unsigned char* hg = s->h_gen;
    SearchParams* h_sp = reinterpret_cast<SearchParams*>(hg);                       hg += 4096;
    float* h_cum = reinterpret_cast<float*>(hg);                                    hg += (size_t)s->rows_cap * 4;
    int* h_rule = reinterpret_cast<int*>(hg);                                       hg += (size_t)s->rows_cap * 16;
    int* h_nsp = reinterpret_cast<int*>(hg);                                        hg += (size_t)s->rows_cap * 4;
    int* h_plen = reinterpret_cast<int*>(hg);                                       hg += (size_t)s->B * 4;
    short* h_anc = reinterpret_cast<short*>(hg);                                    hg += (size_t)s->rows_cap * WLX_T_TEXT * 2 + 192;
    int* h_rows = reinterpret_cast<int*>(hg);                                       hg += ((size_t)s->rows_cap * 4 + s->B) * 4 + 64;
    unsigned char* h_res = s->h_gen + ((hg - s->h_gen + 63) / 64) * 64;
    static_assert(sizeof(SearchParams) <= 4096, "SearchParams staging");
    CKR(fill_search_params(e, s, batch, R, o, apply_ts, h_sp));
    CK(hipMemcpyAsync(s->d_sp, h_sp, sizeof(SearchParams), hipMemcpyHostToDevice, st));
    launch_search_reset(S, batch, rows, st);
    std::vector<int> nsp(rows, 0), pl(batch);
    memset(h_anc, 0, (size_t)rows * WLX_T_TEXT * 2);
    for (int b = 0; b < batch; ++b) {
        pl[b] = plens[b];
        h_plen[b] = plens[b];
        for (int r = 0; r < R; ++r) {
            const int row = b * R + r;
            h_cum[row] = (sampling || r == 0) ? 0.f : -INFINITY;
            short* a = h_anc + (size_t)row * WLX_T_TEXT;
            for (int p = 0; p < pl[b] - 1; ++p) a[p] = (short)(b * R);   // prompt K/V live in the item's first cache row
            a[pl[b] - 1] = (short)row;
            // rule state before the first generated token: no last token, "one before last" counts as a timestamp, no timestamp yet
            h_rule[4 * row] = 0; h_rule[4 * row + 1] = 1; h_rule[4 * row + 2] = -1; h_rule[4 * row + 3] = 0;
        }
    }
    CK(hipMemcpyAsync(S.cum, h_cum, rows * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(S.rule, h_rule, (size_t)rows * 16, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(S.plen, h_plen, batch * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(s->d_anc, h_anc, (size_t)rows * WLX_T_TEXT * sizeof(short), hipMemcpyHostToDevice, st));
    // ---- prefill prompt[0 .. plen-2]; no_speech_prob is read at the sot position
    if (!injected_logits) {
        std::vector<int> sot_at(batch, -1);
        int longest = 0, with_prompt = 0;
        for (int b = 0; b < batch; ++b) {
            const int32_t* pr = prompts + (size_t)b * pstride;
            for (int i = 0; i < pl[b]; ++i) if (pr[i] == o->ids.sot) sot_at[b] = i;
            if (sot_at[b] == pl[b] - 1) { for (int r = 0; r < (sampling ? R : 1); ++r) nsp[b * R + r] = (r == 0) ? 1 : 0; }
            longest = std::max(longest, pl[b] - 1);
            with_prompt += pl[b] > 1 ? 1 : 0;
        }
        // Batched calls with short prompts (the multilingual start sequence `[sot, lang, task]` (+ prefix) of every item of a
        // batch_inference batch): ALL items' prompt rows in ONE decoder pass — item b is row group b (16 rows, the unused ones
        // repeat the item's last prompt row: the same K / V written to the same cache position again) — instead of one full
        // pass per item, each of which streams every decoder weight (large-v3: 1.3 ms per item, 8 items per batch).
        static const bool joint = [] { const char* v = wlx_ab("WLX_PREFILL_JOINT"); return !(v && v[0] == '0'); }();
        if (joint && batch > 1 && with_prompt > 1 && longest <= 16 && s->pf_ok && !s->align && !s->prof && !g_decode_v1) {
            // (the prefill working set holds WLX_T_TEXT rows = 28 items of 16 rows: a wider batch, round 5, goes in blocks of 28 items)
            const int IB = WLX_T_TEXT / 16;
            const Slot::DecBufs& pb = s->pf;
            const int d = e->spec.d_model;
            for (int b0 = 0; b0 < batch; b0 += IB) {
                const int nb = std::min(IB, batch - b0), prow = 16 * nb;
                if ((size_t)(4 * prow + nb) > s->h_stage_ints - 8) return fail(WLX_ERR_ARG, "batch too large");
                CK(hipStreamSynchronize(st));                        // the shared staging may still feed an earlier pass's copies
                int* h = s->h_stage;
                for (int bi = 0; bi < nb; ++bi) {
                    const int b = b0 + bi;
                    const int32_t* pr = prompts + (size_t)b * pstride;
                    const int np_ = pl[b] - 1;                          // prompt rows of this item (0: a lone start token — the group idles on row 0's token at position 0... of a valid cache row)
                    for (int i = 0; i < 16; ++i) {
                        const int j = np_ > 0 ? std::min(i, np_ - 1) : 0;
                        h[bi * 16 + i] = pr[j]; h[prow + bi * 16 + i] = j; h[2 * prow + bi * 16 + i] = b * R; h[3 * prow + bi * 16 + i] = b * R;
                    }
                    h[4 * prow + bi] = enc_items ? enc_items[b] : b;
                }
                CK(hipMemcpyAsync(pb.d_token, h, prow * 4, hipMemcpyHostToDevice, st));
                CK(hipMemcpyAsync(pb.d_pos, h + prow, prow * 4, hipMemcpyHostToDevice, st));
                CK(hipMemcpyAsync(pb.d_cache, h + 2 * prow, prow * 4, hipMemcpyHostToDevice, st));
                CK(hipMemcpyAsync(pb.d_ancrow, h + 3 * prow, prow * 4, hipMemcpyHostToDevice, st));
                CK(hipMemcpyAsync(pb.d_group_item, h + 4 * prow, nb * 4, hipMemcpyHostToDevice, st));
                s->anc_ident = false;
                decoder_pass(e, s, prow, 16, nb, false, false, &pb);
                CK(hipGetLastError());
                for (int bi = 0; bi < nb; ++bi) {
                    const int b = b0 + bi;
                    if (!(sot_at[b] >= 0 && sot_at[b] < pl[b] - 1)) continue;
                    GemvParams p{};
                    p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_F32; p.M = 1; p.K = d; p.KT = d / 32; p.N = V;
                    p.Wp = e->Wvocab; p.bias = nullptr; p.X = pb.xd + (size_t)(bi * 16 + sot_at[b]) * d; p.ldx = d; p.gamma = e->dec_ln_g; p.beta = e->dec_ln_b;
                    p.Y = s->logits; p.ldy = s->ldl; p.qscale = 1.f; p.xsrc = GEMV_X_PLAIN;
                    launch_dec_gemv(p, st);
                    launch_token_prob(s->logits, s->ldl, V, 1, o->ids.no_speech, s->d_tokprob, st);
                    CK(hipMemcpyAsync(S.no_speech + b, s->d_tokprob, 4, hipMemcpyDefault, st));      // (into the pinned result area)
                }
                CK(hipGetLastError());
            }
        } else {
            for (int b = 0; b < batch; ++b) {
                const int32_t* pr = prompts + (size_t)b * pstride;
                if (pl[b] > 1)
                    CKR(prefill_tokens(e, s, enc_items ? enc_items[b] : b, b * R, pr, 0, pl[b] - 1, nullptr, (sot_at[b] >= 0 && sot_at[b] < pl[b] - 1) ? sot_at[b] : -1,
                                       o->ids.no_speech, S.no_speech + b));
            }
        }
    }
    // ---- decode rows
    {
        std::vector<int> tk(rows), ps(rows), ca(rows), an(rows), gi(batch);
        for (int b = 0; b < batch; ++b) {
            gi[b] = enc_items ? enc_items[b] : b;
            for (int r = 0; r < R; ++r) {
                const int row = b * R + r;
                tk[row] = prompts[(size_t)b * pstride + pl[b] - 1]; ps[row] = pl[b] - 1; ca[row] = row; an[row] = row;
            }
        }
        CKR(upload_rows(s, tk, ps, ca, an, gi, h_rows));
        memcpy(h_nsp, nsp.data(), (size_t)rows * 4);
        CK(hipMemcpyAsync(S.nsp_row, h_nsp, rows * 4, hipMemcpyHostToDevice, st));     // (pinned: no wait needed before the loop)
    }
    // ---- autoregressive loop: one graph replay per step. The host never lets the stream run dry: step k+1 is
    // enqueued BEFORE the host knows how step k ended (the update kernels raise a pinned done word and mirror their step
    // number themselves: no copies, no events), so the check costs no GPU idle time; once the word is set the one extra
    // step already in flight only rewrites scratch (its search kernels return at once) and nobody waits for it: the
    // results are read from pinned memory below.
    volatile int* h_done = s->h_stage + (s->h_stage_ints - 4);
    h_done[0] = 0;
    h_done[1] = 0;                                              // the update kernels' step number (search.hip step_mirror)
    const double tg1 = now_us();
    int steps_run = 0;
    bool finished = false;
    const int gsteps = (injected_logits || !e->use_graph) ? 1 : graph_steps_for(s, rows);
    for (int step = 0; step < max_steps && !finished;) {
        int n_this = 1;
        if (injected_logits) {
            if (step >= inj_steps) break;
            // test hook: logits come from the caller; the embed kernel still records the fed tokens
            RowTables rt{s->d_token, s->d_pos, s->d_cache, s->d_ancrow, s->d_anc, s->d_intok};
            launch_dec_embed(e->tok_emb16, e->dec_pos, e->spec.d_model, rt, rows, s->xd, nullptr, st);
            CK(hipMemcpy2DAsync(s->logits, (size_t)s->ldl * 4, inj + (size_t)step * rows * V, (size_t)V * 4, (size_t)V * 4, rows,
                                hipMemcpyHostToDevice, st));
            launch_search(e, s, rows, R, batch, sampling);
            CK(hipGetLastError());
        } else {
            const double ta = gen_trace ? now_us() : 0.0;
            n_this = std::min(gsteps, max_steps - step);            // (an odd step budget ends on a one-step graph)
            CKR(run_step(e, s, rows, R, batch, sampling, n_this));
            if (gen_trace) tg_launch += now_us() - ta;
        }
        steps_run += n_this;
        step += n_this;                                             // = steps enqueued so far
        // The search kernel of the step that finishes the last item stores 1 to the pinned word h_done itself, and every update kernel
        // stores its step number to the pinned word next to it as it ends (search.hip step_mirror): the host reads how far the stream got
        // from that word. Step k+1 is enqueued BEFORE the host waits for step k-1 to have ended, so the stream never runs dry; at most two
        // steps run past the finish (scratch-only, see decoder_pass). Round 5: this replaced a hipEventRecord after every step graph +
        // hipEventSynchronize one step behind, which cost 0.31 ms per 64-step window (profiles/r5p_*: 27.59 vs 27.28 ms with no
        // synchronisation at all; polling every 2nd / 4th step had been measured in round 2 and rejected — a skipped poll lets a whole
        // step run past a real end of text).
        if (injected_logits) {
            CK(hipStreamSynchronize(st));
            finished = h_done[0] != 0;
        } else if (step >= 2) {
            const double ta = gen_trace ? now_us() : 0.0;
            int spins = 0, rounds = 0;
            // wait until all but the LAST enqueued step have ended, i.e. the last one has started: the next launch then lands while it runs
            while (h_done[1] < step - 1 && h_done[0] == 0) {  // update kernel number `step - 1` (1-based) has not ended yet
                // a short pure spin (the word usually moves within one step, 0.1-0.4 ms for one stream), then the core is offered to
                // other runnable threads between polls (every decoding thread — batch lanes, client threads — used to burn a
                // core for the whole generate). No sleep: a timer sleep (>= 50 us of slack) could let a 113 us tiny.en step's stream run dry.
                if (rounds == 0 && spins < 4096) cpu_relax(); else sched_yield();
                if (++spins >= (1 << 16)) {                    // every few ms: is the stream still alive? (a fault must not hang the caller)
                    spins = 0; ++rounds;
                    const hipError_t q = hipStreamQuery(st);
                    if (q == hipSuccess) break;                // drained: the counter is final
                    if (q != hipErrorNotReady) return fail(WLX_ERR_HIP, "decode loop: %s", hipGetErrorString(q));
                }
            }
            if (gen_trace) tg_wait += now_us() - ta;
            finished = h_done[0] != 0;
        }
    }
    const double tg2 = now_us();
    if (gen_trace) CK(hipStreamSynchronize(st));             // (only to tell the drain from the readback in the trace line)
    const double tg3 = now_us();
    // ---- results. Round 6: the update kernels store them in pinned host memory (Slot::h_hyp) and order them before the done word (search.hip
    // finish_item), so a call that saw the word reads them NOW — the step it had already enqueued behind the finish (scratch-only, ~0.4 ms) is still
    // running and is not waited for; the next call on the slot queues behind it on the stream. (Until then six copies were enqueued on the slot stream and
    // one wait covered the tail, the copies and the timing event.) A loop that ended any other way (injected logits, a step budget) waits for the stream.
    CK(hipEventRecord(s->ev1, st));
    s->gen_pending = true;
    if (h_done[0] == 0 || gen_trace) CK(hipStreamSynchronize(st));
    std::atomic_thread_fence(std::memory_order_acquire);
    const int MB = s->max_items;
    const int* n_hyp = S.n_hyp_host;
    const int* hyp_len = S.hyp_len;
    const float* hyp_score = S.hyp_score;
    const float* nspv = S.no_speech;
    const int* hyp_tok = S.hyp_tokens;
    (void)MB; (void)h_res;
    const int NH = std::max(1, o->num_hypotheses);
    for (int b = 0; b < batch; ++b) {
        std::vector<int> order(std::min(n_hyp[b], WLX_MAX_HYP));
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
            return hyp_score[(size_t)b * WLX_MAX_HYP + a] > hyp_score[(size_t)b * WLX_MAX_HYP + c];
        });
        for (int h = 0; h < NH; ++h) {
            int32_t* dst = tokens_out + ((size_t)b * NH + h) * tstride;
            if (h < (int)order.size()) {
                const int src = order[h];
                int len = hyp_len[(size_t)b * WLX_MAX_HYP + src];
                if (len > tstride) len = tstride;
                memcpy(dst, &hyp_tok[((size_t)b * WLX_MAX_HYP + src) * WLX_T_TEXT], (size_t)len * 4);
                n_tokens_out[b * NH + h] = len;
                scores_out[b * NH + h] = hyp_score[(size_t)b * WLX_MAX_HYP + src];
            } else {
                n_tokens_out[b * NH + h] = 0;
                scores_out[b * NH + h] = -INFINITY;
            }
        }
        if (nsp_out) nsp_out[b] = nspv[b];
    }
    if (gen_trace) {
        float gms = 0.f;
        CK(hipEventElapsedTime(&gms, s->ev0, s->ev1));
        fprintf(stderr, "[wlx gen] steps %d: setup %.0f us | loop %.0f us (graph launch calls %.0f, event waits %.0f) | drain %.0f us | readback %.0f us | device %.0f us\n",
                steps_run, tg1 - tg0, tg2 - tg1, tg_launch, tg_wait, tg3 - tg2, now_us() - tg3, 1e3 * gms);
    }
    return WLX_OK;
}

extern "C" int32_t wlx_generate_ex(wlx_engine* e, int32_t slot, int32_t batch, const int32_t* enc_items,
                                   const int32_t* prompts, const int32_t* prompt_lens, int32_t prompt_stride,
                                   const wlx_gen_opts* opts, int32_t* tokens_out, int32_t tokens_stride,
                                   int32_t* n_tokens_out, float* scores_out, float* no_speech_prob_out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (!prompts || !prompt_lens || !tokens_out || !n_tokens_out || !scores_out) return fail(WLX_ERR_ARG, "null argument");
    if (batch < 1 || batch > s->B) return fail(WLX_ERR_ARG, "batch %d out of range (slot max %d)", batch, s->B);
    if (s->enc_batch < 1) return fail(WLX_ERR_STATE, "generate before encode");
    for (int b = 0; b < batch; ++b) {
        const int it = enc_items ? enc_items[b] : b;
        if (it < 0 || it >= s->enc_batch)
            return fail(WLX_ERR_STATE, "generate: item %d uses encoder item %d but only %d are encoded", b, it, s->enc_batch);
    }
    CK(hipSetDevice(e->device));
    return generate_impl(e, s, batch, prompts, prompt_lens, prompt_stride, enc_items, opts, false, nullptr, 0, tokens_out,
                         tokens_stride, n_tokens_out, scores_out, no_speech_prob_out);
}

extern "C" int32_t wlx_generate(wlx_engine* e, int32_t slot, int32_t batch, const int32_t* prompts,
                                const int32_t* prompt_lens, int32_t prompt_stride, const wlx_gen_opts* opts,
                                int32_t* tokens_out, int32_t tokens_stride, int32_t* n_tokens_out, float* scores_out,
                                float* no_speech_prob_out) {
    return wlx_generate_ex(e, slot, batch, nullptr, prompts, prompt_lens, prompt_stride, opts, tokens_out, tokens_stride,
                           n_tokens_out, scores_out, no_speech_prob_out);
}

extern "C" int32_t wlx_debug_search(wlx_engine* e, int32_t slot, const float* logits, int32_t steps,
                                    const int32_t* prompt, int32_t prompt_len, const wlx_gen_opts* opts,
                                    int32_t* tokens_out, int32_t tokens_stride, int32_t* n_tokens_out, float* scores_out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (!logits || !prompt || !opts) return fail(WLX_ERR_ARG, "null argument");
    CK(hipSetDevice(e->device));
    float nsp;
    return generate_impl(e, s, 1, prompt, &prompt_len, prompt_len, nullptr, opts, true, logits, steps, tokens_out, tokens_stride,
                         n_tokens_out, scores_out, &nsp);
}

extern "C" int32_t wlx_detect_language(wlx_engine* e, int32_t slot, int32_t batch, int32_t sot, const int32_t* lang_ids,
                                       int32_t n_lang, float* probs_out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (batch < 1 || batch > s->enc_batch) return fail(WLX_ERR_STATE, "detect_language before encode");
    if (!lang_ids || n_lang < 1 || n_lang > 256 || !probs_out) return fail(WLX_ERR_ARG, "bad language id list");
    if (sot < 0 || sot >= e->spec.vocab) return fail(WLX_ERR_ARG, "bad sot id");
    for (int i = 0; i < n_lang; ++i) if (lang_ids[i] < 0 || lang_ids[i] >= e->spec.vocab) return fail(WLX_ERR_ARG, "bad language id");
    CK(hipSetDevice(e->device));
    hipStream_t st = s->stream;
    // one decoder step on [sot] per item: row b -> cache row b*R (distinct per item)
    std::vector<int> tk(batch, sot), ps(batch, 0), ca(batch), an(batch), gi(batch);
    std::vector<short> anc((size_t)s->cache_rows * WLX_T_TEXT, 0);
    for (int b = 0; b < batch; ++b) { ca[b] = an[b] = b * s->R; gi[b] = b; anc[(size_t)(b * s->R) * WLX_T_TEXT] = (short)(b * s->R); }
    CKR(set_anc_rows(s, anc, 0, s->cache_rows));
    CKR(upload_rows(s, tk, ps, ca, an, gi));
    decoder_pass(e, s, batch, 1, batch, true, false);
    CK(hipMemcpyAsync(s->d_lang_ids, lang_ids, (size_t)n_lang * 4, hipMemcpyHostToDevice, st));
    launch_lang_probs(s->logits, s->ldl, batch, s->d_lang_ids, n_lang, s->d_probs, st);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(probs_out, s->d_probs, (size_t)batch * n_lang * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    return WLX_OK;
}

// ------------------------------------------------------------------------------------------------
// word alignment: ctranslate2 Whisper.align(encoder_output, start_sequence, text_tokens, num_frames, median_filter_width)
// (transcriber_faster_whisper.py:1657-1663; CT2's source is not in the reference tree — the algorithm is the published
// one of openai/whisper timing.py find_alignment / HF generation_whisper.py _extract_token_timestamps):
//   teacher-forced decoder pass over tokens = sot_sequence + [no_timestamps] + text + [eot];
//   text_token_probs[i] = softmax(logits[n_sot + i][: eot])[text[i]];
//   per alignment head: softmax of q.k over the first num_frames/2 encoder positions; (w - mean) / std over the TOKEN axis;
//   median filter of width `median_filter_width` along time (reflect padding); mean over heads;
//   rows n_sot .. n_tokens-2; dynamic time warping on the negated matrix -> monotone (text index, time index) path.
// The decoder pass, the score capture and the token probabilities run on the device; the O(tokens x 1500) scalar
// post-processing (softmax / normalise / median / DTW) runs here on the host side of the call.
static void align_postprocess(const std::vector<float>& scores, int n_heads, int n_tok, int n_sot, int nf, int mw,
                              std::vector<int32_t>& ti, std::vector<int32_t>& fi) {
    const int TP = WLX_T_AUDIO_PAD;
    std::vector<float> w((size_t)n_heads * n_tok * nf);
    for (int h = 0; h < n_heads; ++h)
        for (int t = 0; t < n_tok; ++t) {
            const float* sr = &scores[((size_t)h * n_tok + t) * TP];
            float* wr = &w[((size_t)h * n_tok + t) * nf];
            float mx = sr[0];
            for (int f = 1; f < nf; ++f) mx = std::max(mx, sr[f]);
            double sum = 0.0;
            for (int f = 0; f < nf; ++f) { wr[f] = std::exp(sr[f] - mx); sum += wr[f]; }
            const float inv = (float)(1.0 / sum);
            for (int f = 0; f < nf; ++f) wr[f] *= inv;
        }
    // std / mean over the token axis (population std), per head and frame
    for (int h = 0; h < n_heads; ++h)
        for (int f = 0; f < nf; ++f) {
            double m = 0.0;
            for (int t = 0; t < n_tok; ++t) m += w[((size_t)h * n_tok + t) * nf + f];
            m /= n_tok;
            double v = 0.0;
            for (int t = 0; t < n_tok; ++t) { const double dlt = w[((size_t)h * n_tok + t) * nf + f] - m; v += dlt * dlt; }
            const double sd = std::sqrt(v / n_tok);
            for (int t = 0; t < n_tok; ++t) {
                float& x = w[((size_t)h * n_tok + t) * nf + f];
                x = (float)((x - m) / sd);
            }
        }
    // median filter along time, reflect padding (skipped, as in the reference implementation, when the row is too short)
    const int pad = mw / 2;
    if (mw > 1 && nf > pad) {
        std::vector<float> row(nf + 2 * pad), win(mw);
        for (size_t r = 0; r < (size_t)n_heads * n_tok; ++r) {
            float* wr = &w[r * nf];
            for (int i = 0; i < pad; ++i) { row[i] = wr[pad - i]; row[pad + nf + i] = wr[nf - 2 - i]; }
            std::copy(wr, wr + nf, row.begin() + pad);
            for (int f = 0; f < nf; ++f) {
                std::copy(row.begin() + f, row.begin() + f + mw, win.begin());
                std::nth_element(win.begin(), win.begin() + pad, win.end());
                wr[f] = win[pad];
            }
        }
    }
    // mean over heads, rows n_sot .. n_tok-2, negated: the DTW cost
    const int N = n_tok - 1 - n_sot, M = nf;
    std::vector<float> x((size_t)N * M);
    for (int i = 0; i < N; ++i)
        for (int f = 0; f < M; ++f) {
            float acc = 0.f;
            for (int h = 0; h < n_heads; ++h) acc += w[((size_t)h * n_tok + n_sot + i) * nf + f];
            x[(size_t)i * M + f] = -(acc / (float)n_heads);
        }
    // dynamic time warping (openai/whisper timing.py dtw_cpu)
    const float INF = std::numeric_limits<float>::infinity();
    std::vector<float> cost((size_t)(N + 1) * (M + 1), INF);
    std::vector<int8_t> trace((size_t)(N + 1) * (M + 1), -1);
    cost[0] = 0.f;
    for (int j = 1; j <= M; ++j)
        for (int i = 1; i <= N; ++i) {
            const float c0 = cost[(size_t)(i - 1) * (M + 1) + j - 1], c1 = cost[(size_t)(i - 1) * (M + 1) + j], c2 = cost[(size_t)i * (M + 1) + j - 1];
            float cc; int8_t tt;
            if (c0 < c1 && c0 < c2) { cc = c0; tt = 0; }
            else if (c1 < c0 && c1 < c2) { cc = c1; tt = 1; }
            else { cc = c2; tt = 2; }
            cost[(size_t)i * (M + 1) + j] = x[(size_t)(i - 1) * M + j - 1] + cc;
            trace[(size_t)i * (M + 1) + j] = tt;
        }
    for (int j = 0; j <= M; ++j) trace[j] = 2;
    for (int i = 0; i <= N; ++i) trace[(size_t)i * (M + 1)] = 1;
    int i = N, j = M;
    ti.clear(); fi.clear();
    while (i > 0 || j > 0) {
        ti.push_back(i - 1); fi.push_back(j - 1);
        const int8_t tt = trace[(size_t)i * (M + 1) + j];
        if (tt == 0) { --i; --j; } else if (tt == 1) --i; else --j;
    }
    std::reverse(ti.begin(), ti.end());
    std::reverse(fi.begin(), fi.end());
}

extern "C" int32_t wlx_align(wlx_engine* e, int32_t slot, int32_t item, const int32_t* tokens, int32_t n_tokens, int32_t n_sot,
                             int32_t num_frames, int32_t median_filter_width, const int32_t* heads, int32_t n_heads, int32_t eot,
                             int32_t* text_indices, int32_t* time_indices, int32_t path_cap, int32_t* n_path_out,
                             float* text_token_probs) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (!tokens || !heads || !text_indices || !time_indices || !n_path_out || !text_token_probs) return fail(WLX_ERR_ARG, "null argument");
    if (item < 0 || item >= s->enc_batch) return fail(WLX_ERR_STATE, "align: item %d not encoded", item);
    if (n_sot < 1 || n_tokens < n_sot + 3 || n_tokens > WLX_T_TEXT) return fail(WLX_ERR_ARG, "align: %d tokens with a start sequence of %d", n_tokens, n_sot);
    if (n_heads < 1 || n_heads > e->spec.dec_layers * e->H) return fail(WLX_ERR_ARG, "align: bad head count");
    for (int i = 0; i < n_heads; ++i)
        if (heads[2 * i] < 0 || heads[2 * i] >= e->spec.dec_layers || heads[2 * i + 1] < 0 || heads[2 * i + 1] >= e->H)
            return fail(WLX_ERR_ARG, "align: head (%d, %d) out of range", heads[2 * i], heads[2 * i + 1]);
    for (int i = 0; i < n_tokens; ++i) if (tokens[i] < 0 || tokens[i] >= e->spec.vocab) return fail(WLX_ERR_ARG, "align: token out of vocabulary");
    if (eot < 1 || eot > e->spec.vocab || median_filter_width < 1 || (median_filter_width & 1) == 0) return fail(WLX_ERR_ARG, "align: bad eot / filter width");
    int nf = num_frames / 2;
    if (nf < 1) nf = 1;
    if (nf > WLX_T_AUDIO) nf = WLX_T_AUDIO;
    const int n_text = n_tokens - n_sot - 2;
    CK(hipSetDevice(e->device));
    hipStream_t st = s->stream;
    const size_t need = (size_t)n_heads * n_tokens * WLX_T_AUDIO_PAD;
    if (need > s->align_cap) {
        CK(hipStreamSynchronize(st));
        if (s->align_scores) CK(hipFree(s->align_scores));
        s->align_scores = nullptr; s->align_cap = 0;
        CK(hipMalloc(reinterpret_cast<void**>(&s->align_scores), need * sizeof(float)));
        s->align_cap = need;
    }
    const int crow = item * s->R;
    std::vector<short> anc((size_t)WLX_T_TEXT, (short)crow);
    CKR(set_anc_rows(s, anc, crow, 1));
    Slot::AlignCapture cap{s->align_scores, heads, n_heads, n_tokens, 0, item};
    std::vector<float> probs(n_tokens, 0.f);
    int rc = WLX_OK;
    s->align = &cap;
    for (int c0 = 0; c0 < n_tokens && rc == WLX_OK; c0 += 64) {
        const int rows = std::min(64, n_tokens - c0);
        const int groups = (rows + 15) / 16;
        std::vector<int> tk(rows), ps(rows), ca(rows, crow), an(rows, crow), gi(groups, item), tgt(rows, -1);
        for (int i = 0; i < rows; ++i) {
            tk[i] = tokens[c0 + i]; ps[i] = c0 + i;
            const int p = c0 + i;                       // logits at position p predict tokens[p + 1]
            if (p >= n_sot && p < n_sot + n_text) tgt[i] = tokens[p + 1];
        }
        rc = upload_rows(s, tk, ps, ca, an, gi);
        if (rc != WLX_OK) break;
        cap.row0 = c0;
        decoder_pass(e, s, rows, 16, groups, true, false);
        if (hipMemcpyAsync(s->d_align_tgt, tgt.data(), (size_t)rows * 4, hipMemcpyHostToDevice, st) != hipSuccess) { rc = WLX_ERR_HIP; break; }
        launch_token_prob_rows(s->logits, s->ldl, eot, rows, s->d_align_tgt, s->d_align_prob, st);
        if (hipMemcpyAsync(probs.data() + c0, s->d_align_prob, (size_t)rows * 4, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = WLX_ERR_HIP; break; }
        if (hipStreamSynchronize(st) != hipSuccess) { rc = WLX_ERR_HIP; break; }   // tgt / staging reuse
    }
    s->align = nullptr;
    if (rc != WLX_OK) return fail(rc, "align: decoder pass failed");
    CK(hipGetLastError());
    std::vector<float> scores(need);
    CK(hipMemcpyAsync(scores.data(), s->align_scores, need * sizeof(float), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    std::vector<int32_t> ti, fi;
    align_postprocess(scores, n_heads, n_tokens, n_sot, nf, median_filter_width, ti, fi);
    if ((int)ti.size() > path_cap) return fail(WLX_ERR_ARG, "align: path of %d steps exceeds the caller's capacity %d", (int)ti.size(), path_cap);
    memcpy(text_indices, ti.data(), ti.size() * 4);
    memcpy(time_indices, fi.data(), fi.size() * 4);
    *n_path_out = (int32_t)ti.size();
    for (int i = 0; i < n_text; ++i) text_token_probs[i] = probs[n_sot + i];
    return WLX_OK;
}

// ------------------------------------------------------------------------------------------------
// test hooks
extern "C" int32_t wlx_debug_logits_get(wlx_engine* e, int32_t slot, float* out, int32_t rows, int64_t cap_floats) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    const int V = e->spec.vocab;
    if (!out || rows < 1 || rows > s->rows_cap || (int64_t)rows * V > cap_floats) return fail(WLX_ERR_ARG, "bad rows/cap");
    CK(hipSetDevice(e->device));
    CK(hipMemcpy2DAsync(out, (size_t)V * 4, s->logits, (size_t)s->ldl * 4, (size_t)V * 4, rows, hipMemcpyDeviceToHost, s->stream));
    CK(hipStreamSynchronize(s->stream));
    return WLX_OK;
}

extern "C" int32_t wlx_debug_decode_logits(wlx_engine* e, int32_t slot, const int32_t* tokens, int32_t n, float* out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (!tokens || !out || n < 1 || n > WLX_T_TEXT) return fail(WLX_ERR_ARG, "bad tokens");
    if (s->enc_batch < 1) return fail(WLX_ERR_STATE, "decode before encode");
    for (int i = 0; i < n; ++i) if (tokens[i] < 0 || tokens[i] >= e->spec.vocab) return fail(WLX_ERR_ARG, "token out of vocabulary");
    CK(hipSetDevice(e->device));
    std::vector<short> anc(WLX_T_TEXT, 0);   // cache row 0, identity ancestry
    CKR(set_anc_rows(s, anc, 0, 1));
    CKR(prefill_tokens(e, s, 0, 0, tokens, 0, n, out, -1, 0, nullptr));
    CK(hipStreamSynchronize(s->stream));
    return WLX_OK;
}

extern "C" int32_t wlx_debug_time_decode_step(wlx_engine* e, int32_t slot, int32_t rows, int32_t t, int32_t iters,
                                              float* avg_ms_out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (rows < 1 || rows > s->cache_rows || rows > s->rows_cap || t < 0 || t >= WLX_T_TEXT || iters < 1 || !avg_ms_out)
        return fail(WLX_ERR_ARG, "bad arguments");
    if (rows > 16 && (rows % s->R != 0 || rows / s->R > s->B)) return fail(WLX_ERR_ARG, "more than 16 rows: a multiple of the slot's rows per item");
    if (s->enc_batch < 1) return fail(WLX_ERR_STATE, "decode before encode");
    s->busy_variant = device_is_busy(s);     // (the launch shapes a step captured now would use)
    CK(hipSetDevice(e->device));
    hipStream_t st = s->stream;
    // one item, `rows` beams all at position t with identity history (timing only: cache content is whatever is there);
    // more than 16 rows: rows / R items of R beams each, as a batched decode has them
    const int tR = rows > 16 ? s->R : rows, tG = rows / tR;
    std::vector<int> tk(rows, 0), ps(rows, t), ca(rows), an(rows), gi(tG, 0);
    for (int g = 0; g < tG; ++g) gi[g] = g % std::max(1, s->enc_batch);
    std::vector<short> anc((size_t)rows * WLX_T_TEXT);
    for (int r = 0; r < rows; ++r) { ca[r] = an[r] = r; for (int p = 0; p < WLX_T_TEXT; ++p) anc[(size_t)r * WLX_T_TEXT + p] = (short)r; }
    CKR(set_anc_rows(s, anc, 0, rows));
    CKR(upload_rows(s, tk, ps, ca, an, gi));
    CK(hipMemsetAsync(s->st.done, 0, 4, st));
    hipGraph_t graph; hipGraphExec_t exec;
    decoder_pass(e, s, rows, tR, tG, true, true);      // eager first (dynamic-LDS limits are raised outside capture)
    constexpr int passes = 1;      // (several steps per graph were measured in round 2: the ~7 us graph-to-graph boundary is not worth running past a transcript's end)
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int q = 0; q < passes; ++q) decoder_pass(e, s, rows, tR, tG, true, true);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphDestroy(graph));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(s->ev0, st));
    for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(s->ev1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *avg_ms_out = ms / (float)iters / (float)passes;
    CK(hipGraphExecDestroy(exec));
    return WLX_OK;
}

// In-kernel timeline of ONE decode step (scripts/trace_step.py). Only libwlx_trace.so (-DWLX_TRACE) records anything;
// the production library reports WLX_ERR_STATE. out: [n_launches][WLX_TR_STRIDE] u64 records, names: [n_launches][48].
extern "C" int32_t wlx_debug_trace_step(wlx_engine* e, int32_t slot, int32_t rows, int32_t t, int32_t with_search,
                                        uint64_t* out, int64_t cap_u64, char* names, int32_t* n_launches_out) {
#ifndef WLX_TRACE
    (void)e; (void)slot; (void)rows; (void)t; (void)with_search; (void)out; (void)cap_u64; (void)names; (void)n_launches_out;
    return fail(WLX_ERR_STATE, "libwlx.so was built without -DWLX_TRACE (use libwlx_trace.so, scripts/trace_step.py)");
#else
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (rows < 1 || rows > s->cache_rows || rows > s->rows_cap || t < 0 || t >= WLX_T_TEXT || !out || !names || !n_launches_out)
        return fail(WLX_ERR_ARG, "bad arguments");
    if (rows > 16 && (rows % s->R != 0 || rows / s->R > s->B)) return fail(WLX_ERR_ARG, "more than 16 rows: a multiple of the slot's rows per item");
    if (s->enc_batch < 1) return fail(WLX_ERR_STATE, "decode before encode");
    CK(hipSetDevice(e->device));
    hipStream_t st = s->stream;
    const int tR = rows > 16 ? s->R : rows, tG = rows / tR;
    std::vector<int> tk(rows, 0), ps(rows, t), ca(rows), an(rows), gi(tG, 0);
    for (int g = 0; g < tG; ++g) gi[g] = g % std::max(1, s->enc_batch);
    std::vector<short> anc((size_t)rows * WLX_T_TEXT);
    for (int r = 0; r < rows; ++r) { ca[r] = an[r] = r; for (int p = 0; p < WLX_T_TEXT; ++p) anc[(size_t)r * WLX_T_TEXT + p] = (short)r; }
    const size_t max_launch = 320;
    unsigned long long* buf = nullptr;
    CK(hipMalloc(&buf, max_launch * WLX_TR_STRIDE * 8));
    g_trace_buf = buf; g_trace_seq = 0;
    hipGraph_t graph; hipGraphExec_t exec;
    auto reset_state = [&]() -> int {
        CKR(set_anc_rows(s, anc, 0, rows));
        CKR(upload_rows(s, tk, ps, ca, an, gi));
        CK(hipMemsetAsync(s->st.done, 0, 4, st)); CK(hipMemsetAsync(s->st.item_done, 0, 4, st));
        CK(hipMemsetAsync(s->st.n_hyp, 0, 4, st)); CK(hipMemsetAsync(s->st.n_finished, 0, 4, st));
        static const int rule0[16 * 4] = {0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0,
                                          0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0, 0, 1, -1, 0};
        CK(hipMemcpyAsync(s->st.rule, rule0, (size_t)rows * 16, hipMemcpyHostToDevice, st));
        return WLX_OK;
    };
    CKR(reset_state());
    { const int seq0 = g_trace_seq; unsigned long long* b0 = g_trace_buf; g_trace_buf = nullptr;   // eager pass (LDS limits), untraced
      decoder_pass(e, s, rows, tR, tG, true, true); g_trace_seq = seq0; g_trace_buf = b0; }
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    decoder_pass(e, s, rows, tR, tG, true, true);
    if (with_search) launch_search(e, s, rows, tR, tG, false);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphDestroy(graph));
    const int n = g_trace_seq;
    g_trace_buf = nullptr;
    if (n > (int)max_launch) { (void)hipGraphExecDestroy(exec); (void)hipFree(buf); return fail(WLX_ERR_ARG, "trace: %d launches exceed the trace buffer (%d)", n, (int)max_launch); }
    for (int i = 0; i < 3; ++i) { CKR(reset_state()); CK(hipGraphLaunch(exec, st)); }
    CKR(reset_state());
    CK(hipMemsetAsync(buf, 0, max_launch * WLX_TR_STRIDE * 8, st));
    CK(hipStreamSynchronize(st));
    CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    CK(hipGraphExecDestroy(exec));
    if ((int64_t)n * WLX_TR_STRIDE > cap_u64) { (void)hipFree(buf); return fail(WLX_ERR_ARG, "trace buffer too small"); }
    CK(hipMemcpyAsync(out, buf, (size_t)n * WLX_TR_STRIDE * 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    CK(hipFree(buf));
    for (int i = 0; i < n; ++i) { strncpy(names + (size_t)i * 48, g_trace_names[i] ? g_trace_names[i] : "?", 47); names[(size_t)i * 48 + 47] = 0; }
    *n_launches_out = n;
    return WLX_OK;
#endif
}

extern "C" int32_t wlx_debug_profile_step(wlx_engine* e, int32_t slot, int32_t rows, int32_t t, int32_t iters,
                                          wlx_kernel_stat* out, int32_t cap, int32_t* n_out) {
    SlotGuard sg_;
    CKR(slot_acquire(e, slot, sg_));
    Slot* s = sg_.s;
    if (rows < 1 || rows > s->cache_rows || rows > s->rows_cap || t < 0 || t >= WLX_T_TEXT || iters < 1 || !out || !n_out || cap < 1)
        return fail(WLX_ERR_ARG, "bad arguments");
    if (rows > 16 && (rows % s->R != 0 || rows / s->R > s->B)) return fail(WLX_ERR_ARG, "more than 16 rows: a multiple of the slot's rows per item");
    if (s->enc_batch < 1) return fail(WLX_ERR_STATE, "decode before encode");
    s->busy_variant = device_is_busy(s);     // (the launch shapes a step captured now would use)
    CK(hipSetDevice(e->device));
    hipStream_t st = s->stream;
    // up to 16 rows: one item with `rows` beams; more: rows / R items of R beams each, as a batched decode has them
    const int tR = rows > 16 ? s->R : rows, tG = rows / tR;
    std::vector<int> tk(rows, 0), ps(rows, t), ca(rows), an(rows), gi(tG, 0);
    for (int g = 0; g < tG; ++g) gi[g] = g % std::max(1, s->enc_batch);
    std::vector<short> anc((size_t)rows * WLX_T_TEXT);
    for (int r = 0; r < rows; ++r) { ca[r] = an[r] = r; for (int p = 0; p < WLX_T_TEXT; ++p) anc[(size_t)r * WLX_T_TEXT + p] = (short)r; }
    CKR(set_anc_rows(s, anc, 0, rows));
    CKR(upload_rows(s, tk, ps, ca, an, gi));
    CK(hipMemsetAsync(s->st.done, 0, 4, st));
    for (int i = 0; i < 2; ++i) decoder_pass(e, s, rows, tR, tG, true, true);   // warm caches / code objects
    CK(hipStreamSynchronize(st));
    Prof prof;
    prof.t = t;
    s->prof = &prof;
    decoder_pass(e, s, rows, tR, tG, true, true);                               // pass 1: list the launches of one step
    struct Agg { int launches = 0; double bytes = 0, us = 0; };
    std::map<std::string, Agg> agg;
    for (auto& r : prof.recs) { Agg& a = agg[r.name]; a.launches += 1; a.bytes += r.bytes; }
    // pass 2, per kernel name: a graph with just that kernel's launches of the step (back to back on the slot stream, so
    // each pays the dependent-launch boundary exactly as inside the real step), replayed `iters` times between one
    // HIP-event pair. An event pair around every single 2-5 us launch measured the events, not the kernels.
    prof.list_only = false;
    int rc = WLX_OK;
    for (auto& kv : agg) {
        prof.only = kv.first;
        hipGraph_t graph; hipGraphExec_t exec;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) { rc = WLX_ERR_HIP; break; }
        decoder_pass(e, s, rows, tR, tG, true, true);
        if (hipStreamEndCapture(st, &graph) != hipSuccess) { rc = WLX_ERR_HIP; break; }
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(graph); rc = WLX_ERR_HIP; break; }
        (void)hipGraphDestroy(graph);
        for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(exec, st);
        (void)hipEventRecord(s->ev0, st);
        for (int i = 0; i < iters; ++i) (void)hipGraphLaunch(exec, st);
        (void)hipEventRecord(s->ev1, st);
        (void)hipStreamSynchronize(st);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, s->ev0, s->ev1);
        (void)hipGraphExecDestroy(exec);
        kv.second.us = 1000.0 * ms / iters;                                       // all launches of this kernel in one step
    }
    s->prof = nullptr;
    if (rc != WLX_OK) return fail(rc, "profile capture failed");
    CK(hipGetLastError());
    int n = 0;
    for (auto& kv : agg) {
        if (n >= cap) break;
        wlx_kernel_stat& o = out[n++];
        memset(&o, 0, sizeof(o));
        snprintf(o.name, sizeof(o.name), "%s", kv.first.c_str());
        o.launches_per_step = (float)kv.second.launches;
        o.avg_us = (float)(kv.second.us / kv.second.launches);
        o.total_us_per_step = (float)kv.second.us;
        o.bytes_per_launch = kv.second.bytes / kv.second.launches;
    }
    *n_out = n;
    return WLX_OK;
}
