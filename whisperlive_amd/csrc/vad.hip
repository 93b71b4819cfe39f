// vad.hip — Silero-VAD speech probabilities on gfx950: the network the reference
// reaches through faster_whisper.vad.get_speech_timestamps -> onnxruntime, one CPU thread
// (whisper_live/transcriber/transcriber_faster_whisper.py:830-838, whisper_live/batch_inference.py:245-248; I/O contract
// whisper_live/vad.py:50-109). CPU restatement: oracle/silero_vad.py.
//
// Shape of the work, 30 s chunk = 938 windows of 512 samples (+64 context):
//   front end (independent per window): STFT-as-conv 258x256 on 4 frames -> |.| -> 4 x (conv1d k3 + ReLU) -> feature[128]
//     -> input half of the LSTM gates gx[512] = W_ih f + b_ih + b_hh.      ~1.1 MFLOP and 2.3 KB of PCM per window against
//     0.7 MB of fp32 weights that stay in L2: arithmetic on the vector ALUs, all activations of a window group in LDS,
//     one workgroup per VAD_WT windows (235 workgroups for 30 s: fills the chip once).
//   recurrence (sequential over windows): g = gx[t] + W_hh h; (i,f,g,o) -> c,h.   ONE workgroup: W_hh (256 KB fp32)
//     lives in the registers of 512 lanes for the whole sequence — lane (unit j, quarter q) holds the four gate rows of
//     unit j over 32 of the 128 inputs — so a step is 8 broadcast LDS reads of h, 64 packed FMAs, a 2-step quad DPP
//     reduction, the gate non-linearities and one barrier. It is latency-bound by construction (938 dependent steps);
//     the CPU path it replaces is the same chain at ~30-50 us per step.
//   output: p[t] = sigmoid(w_out . relu(h_t) + b), parallel over windows.
// fp32 throughout (the gate thresholds at 0.5 / 0.35 are compared against these numbers; the oracle's own fp32-vs-fp64
// difference is 5e-7). Everything is deterministic: fixed summation orders, no atomics.
#include "engine.h"
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace wlx {

constexpr int VAD_WINDOW = 512, VAD_CTX = 64, VAD_NFFT = 256, VAD_HOP = 128, VAD_BINS = 129, VAD_H = 128;
constexpr int VAD_XLEN = VAD_CTX + VAD_WINDOW + VAD_CTX;   // 640: context + window + reflected tail
constexpr int VAD_BSTRIDE = 260;                             // basisT row: 129 real + 129 imaginary filters, padded
constexpr int VAD_WT = 4;                                    // windows per front-end workgroup
constexpr int VAD_FE_THREADS = 256;

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct VadW {            // device pointers, kernel layouts (see vad_repack)
    const float* basisT;   // [256 n][260]: [n][k] real filter k, [n][129 + k] imaginary filter k
    const float* w1T;      // [129 ci][3][128 co]
    const float* w2T;      // [128 ci][3][64 co]
    const float* w3T;      // [64 ci][2][64 co]   taps 1, 2 (tap 0 only ever meets the left zero pad)
    const float* w4T;      // [64 ci][128 co]      centre tap (input length 1)
    const float* b1; const float* b2; const float* b3; const float* b4;
    const float* wihT;     // [128 k][512 r]
    const float* bias;     // [512] b_ih + b_hh
    const float* whhP;     // [4 gate][32 e][512 lane]: W_hh[gate*128 + j][32q + e], lane = 4j + q
    const float* out_w;    // [128]
    float out_b;
};

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// ---------------------------------------------------------------------------------------------------------------------
// front end: VAD_WT windows per workgroup, PCM -> gx
__global__ __launch_bounds__(VAD_FE_THREADS) void vad_frontend_kernel(const float* __restrict__ pcm, long long n, int n_windows,
                                                                       VadW W, float* __restrict__ gx) {
    __shared__ __attribute__((aligned(16))) float xs[VAD_WT][VAD_XLEN];
    __shared__ __attribute__((aligned(16))) float mag[VAD_WT][VAD_BINS][6];    // positions -1..4 (zero pads at both ends)
    __shared__ __attribute__((aligned(16))) float y1[VAD_WT][128][6];
    __shared__ __attribute__((aligned(16))) float y2[VAD_WT][64][4];           // positions -1..1 (+1 unused)
    __shared__ float y3[VAD_WT][64];
    __shared__ float y4[VAD_WT][VAD_H];
    __shared__ float nyq[16][VAD_WT * 4][2];
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * VAD_WT;

    // ---- A: context + window (zeros before the start and past the end), then the reflected tail
    for (int i = tid; i < VAD_WT * (VAD_CTX + VAD_WINDOW); i += VAD_FE_THREADS) {
        const int w = i / (VAD_CTX + VAD_WINDOW), o = i - w * (VAD_CTX + VAD_WINDOW);
        const long long src = (long long)(w0 + w) * VAD_WINDOW - VAD_CTX + o;
        xs[w][o] = (src >= 0 && src < n && w0 + w < n_windows) ? pcm[src] : 0.0f;
    }
    __syncthreads();
    {
        const int w = tid >> 6, j = tid & 63;                                   // 256 threads = 4 windows x 64
        xs[w][VAD_CTX + VAD_WINDOW + j] = xs[w][VAD_CTX + VAD_WINDOW - 2 - j];
    }
    __syncthreads();

    // ---- B: STFT magnitude. Bins 0..127: lane (k, half) does 8 frames (2 windows); the Nyquist bin is split over n.
    {
        const int k = tid & 127, half = tid >> 7;
        float re[8], im[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) re[f] = im[f] = 0.0f;
        const float* bp = W.basisT + k;
        for (int nn = 0; nn < VAD_NFFT; nn += 4) {
            float br[4], bi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { br[u] = bp[(nn + u) * VAD_BSTRIDE]; bi[u] = bp[(nn + u) * VAD_BSTRIDE + VAD_BINS]; }
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(&xs[half * 2 + (f >> 2)][(f & 3) * VAD_HOP + nn]);
#pragma unroll
                for (int u = 0; u < 4; ++u) { re[f] = fmaf(br[u], s[u], re[f]); im[f] = fmaf(bi[u], s[u], im[f]); }
            }
        }
#pragma unroll
        for (int f = 0; f < 8; ++f) mag[half * 2 + (f >> 2)][k][1 + (f & 3)] = sqrtf(re[f] * re[f] + im[f] * im[f]);
        // Nyquist bin (k = 128): lane (frame f, part p) sums 16 taps; 16 parts are added in fixed order below
        const int f = tid & 15, p = tid >> 4;
        float nr = 0.0f, ni = 0.0f;
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
            const int nn = p * 16 + u;
            const float s = xs[f >> 2][(f & 3) * VAD_HOP + nn];
            nr = fmaf(W.basisT[nn * VAD_BSTRIDE + 128], s, nr);
            ni = fmaf(W.basisT[nn * VAD_BSTRIDE + VAD_BINS + 128], s, ni);
        }
        nyq[p][f][0] = nr; nyq[p][f][1] = ni;
        for (int i = tid; i < VAD_WT * VAD_BINS; i += VAD_FE_THREADS) {          // zero pads of mag
            mag[i / VAD_BINS][i % VAD_BINS][0] = 0.0f; mag[i / VAD_BINS][i % VAD_BINS][5] = 0.0f;
        }
    }
    __syncthreads();
    if (tid < VAD_WT * 4) {
        float nr = 0.0f, ni = 0.0f;
        for (int p = 0; p < 16; ++p) { nr += nyq[p][tid][0]; ni += nyq[p][tid][1]; }
        mag[tid >> 2][128][1 + (tid & 3)] = sqrtf(nr * nr + ni * ni);
    }
    __syncthreads();

    // ---- C: conv 129 -> 128, stride 1: lane (co, window pair) -> 2 windows x 4 positions
    {
        const int co = tid & 127, wh = tid >> 7;
        float acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[a][t] = 0.0f;
        for (int ci = 0; ci < VAD_BINS; ++ci) {
            const float wa = W.w1T[(ci * 3 + 0) * 128 + co], wb = W.w1T[(ci * 3 + 1) * 128 + co], wc = W.w1T[(ci * 3 + 2) * 128 + co];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float* m = mag[wh * 2 + a][ci];
                const f32x2 m01 = *reinterpret_cast<const f32x2*>(m), m23 = *reinterpret_cast<const f32x2*>(m + 2),
                            m45 = *reinterpret_cast<const f32x2*>(m + 4);
                acc[a][0] = fmaf(wc, m23[0], fmaf(wb, m01[1], fmaf(wa, m01[0], acc[a][0])));
                acc[a][1] = fmaf(wc, m23[1], fmaf(wb, m23[0], fmaf(wa, m01[1], acc[a][1])));
                acc[a][2] = fmaf(wc, m45[0], fmaf(wb, m23[1], fmaf(wa, m23[0], acc[a][2])));
                acc[a][3] = fmaf(wc, m45[1], fmaf(wb, m45[0], fmaf(wa, m23[1], acc[a][3])));
            }
        }
        const float b = W.b1[co];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float* y = y1[wh * 2 + a][co];
            y[0] = 0.0f; y[5] = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) y[1 + t] = fmaxf(acc[a][t] + b, 0.0f);
        }
    }
    __syncthreads();

    // ---- D: conv 128 -> 64, stride 2 (4 -> 2 positions): lane (co, window)
    {
        const int co = tid & 63, w = tid >> 6;
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll 4
        for (int ci = 0; ci < 128; ++ci) {
            const float wa = W.w2T[(ci * 3 + 0) * 64 + co], wb = W.w2T[(ci * 3 + 1) * 64 + co], wc = W.w2T[(ci * 3 + 2) * 64 + co];
            const float* y = y1[w][ci];
            const f32x2 p01 = *reinterpret_cast<const f32x2*>(y), p23 = *reinterpret_cast<const f32x2*>(y + 2);
            const float p4 = y[4];
            a0 = fmaf(wc, p23[0], fmaf(wb, p01[1], fmaf(wa, p01[0], a0)));      // positions -1, 0, 1
            a1 = fmaf(wc, p4, fmaf(wb, p23[1], fmaf(wa, p23[0], a1)));          // positions  1, 2, 3
        }
        const float b = W.b2[co];
        y2[w][co][0] = 0.0f;
        y2[w][co][1] = fmaxf(a0 + b, 0.0f);
        y2[w][co][2] = fmaxf(a1 + b, 0.0f);
    }
    __syncthreads();

    // ---- E: conv 64 -> 64, stride 2 (2 -> 1 position; taps 1, 2 meet positions 0, 1)
    {
        const int co = tid & 63, w = tid >> 6;
        float a = 0.0f;
#pragma unroll 4
        for (int ci = 0; ci < 64; ++ci)
            a = fmaf(W.w3T[(ci * 2 + 1) * 64 + co], y2[w][ci][2], fmaf(W.w3T[(ci * 2 + 0) * 64 + co], y2[w][ci][1], a));
        y3[w][co] = fmaxf(a + W.b3[co], 0.0f);
    }
    __syncthreads();

    // ---- F: conv 64 -> 128 on one position (centre tap)
    {
        const int co = tid & 127, wh = tid >> 7;
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll 4
        for (int ci = 0; ci < 64; ++ci) {
            const float wv = W.w4T[ci * 128 + co];
            a0 = fmaf(wv, y3[wh * 2][ci], a0);
            a1 = fmaf(wv, y3[wh * 2 + 1][ci], a1);
        }
        const float b = W.b4[co];
        y4[wh * 2][co] = fmaxf(a0 + b, 0.0f);
        y4[wh * 2 + 1][co] = fmaxf(a1 + b, 0.0f);
    }
    __syncthreads();

    // ---- G: input half of the gates: rows r = tid, tid + 256 for the 4 windows
    {
        float acc[2][VAD_WT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int w = 0; w < VAD_WT; ++w) acc[a][w] = 0.0f;
#pragma unroll 4
        for (int k = 0; k < VAD_H; ++k) {
            const float wa = W.wihT[k * 512 + tid], wb = W.wihT[k * 512 + 256 + tid];
#pragma unroll
            for (int w = 0; w < VAD_WT; ++w) {
                const float f = y4[w][k];
                acc[0][w] = fmaf(wa, f, acc[0][w]);
                acc[1][w] = fmaf(wb, f, acc[1][w]);
            }
        }
        const float ba = W.bias[tid], bb = W.bias[256 + tid];
#pragma unroll
        for (int w = 0; w < VAD_WT; ++w)
            if (w0 + w < n_windows) {
                gx[(long long)(w0 + w) * 512 + tid] = acc[0][w] + ba;
                gx[(long long)(w0 + w) * 512 + 256 + tid] = acc[1][w] + bb;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// recurrence: one workgroup, W_hh register-resident
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {      // DPP quad_perm move: CTRL = p0 | p1 << 2 | p2 << 4 | p3 << 6
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

__global__ __launch_bounds__(512) void vad_lstm_kernel(const float* __restrict__ gx, const float* __restrict__ whhP,
                                                        float* __restrict__ hs, int T) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][VAD_H];
    const int tid = threadIdx.x, j = tid >> 2, q = tid & 3;
    f32x2 wi[16], wf[16], wg[16], wo[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        wi[e] = f32x2{whhP[((0 * 32 + 2 * e) * 512) + tid], whhP[((0 * 32 + 2 * e + 1) * 512) + tid]};
        wf[e] = f32x2{whhP[((1 * 32 + 2 * e) * 512) + tid], whhP[((1 * 32 + 2 * e + 1) * 512) + tid]};
        wg[e] = f32x2{whhP[((2 * 32 + 2 * e) * 512) + tid], whhP[((2 * 32 + 2 * e + 1) * 512) + tid]};
        wo[e] = f32x2{whhP[((3 * 32 + 2 * e) * 512) + tid], whhP[((3 * 32 + 2 * e + 1) * 512) + tid]};
    }
    if (tid < VAD_H) hbuf[0][tid] = 0.0f;
    float c = 0.0f;
    float gxv = T > 0 ? gx[q * VAD_H + j] : 0.0f;          // lane q brings in gate q's input half for unit j
    {   // consume every preloaded register once before the loop: the compiler then settles all pending loads HERE instead
        // of carrying a conservative vmcnt wait into the loop header, where it would also drain the previous step's store
        f32x2 chk = {gxv, 0.0f};
#pragma unroll
        for (int e = 0; e < 16; ++e) chk += wi[e] + wf[e] + wg[e] + wo[e];
        if (chk[0] + chk[1] == 1.2345e38f && T < 0) hs[tid] = chk[0];
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const float gnext = gx[(long long)min(t + 1, T - 1) * 512 + q * VAD_H + j];      // branch-free prefetch
        const f32x4* hp = reinterpret_cast<const f32x4*>(&hbuf[t & 1][32 * q]);
        f32x2 ai = {0.0f, 0.0f}, af = {0.0f, 0.0f}, ag = {0.0f, 0.0f}, ao = {0.0f, 0.0f};
#pragma unroll
        for (int e4 = 0; e4 < 8; ++e4) {
            const f32x4 h4 = hp[e4];
            const f32x2 lo = {h4[0], h4[1]}, hi = {h4[2], h4[3]};
            ai = __builtin_elementwise_fma(wi[2 * e4], lo, ai); ai = __builtin_elementwise_fma(wi[2 * e4 + 1], hi, ai);
            af = __builtin_elementwise_fma(wf[2 * e4], lo, af); af = __builtin_elementwise_fma(wf[2 * e4 + 1], hi, af);
            ag = __builtin_elementwise_fma(wg[2 * e4], lo, ag); ag = __builtin_elementwise_fma(wg[2 * e4 + 1], hi, ag);
            ao = __builtin_elementwise_fma(wo[2 * e4], lo, ao); ao = __builtin_elementwise_fma(wo[2 * e4 + 1], hi, ao);
        }
        // Transpose-reduce over the quad: every lane holds a partial of all four gates over its quarter of h; after two
        // exchange steps lane q holds the complete pre-activation of gate q (3 DPP moves instead of an 8-move butterfly),
        // applies that gate's non-linearity once (tanh(x) = 2 sigmoid(2x) - 1 keeps the code uniform), and the four
        // activations are broadcast back: 4 transcendental instructions per lane and step instead of 10.
        const float p0 = ai[0] + ai[1], p1 = af[0] + af[1], p2 = ag[0] + ag[1], p3 = ao[0] + ao[1];
        const bool b0 = q & 1, b1 = q & 2;
        const float xa = (b0 ? p1 : p0) + quad_perm<0xB1>(b0 ? p0 : p1);        // gate b0      over lanes {q, q^1}
        const float xb = (b0 ? p3 : p2) + quad_perm<0xB1>(b0 ? p2 : p3);        // gate 2 + b0
        const float s = (b1 ? xb : xa) + quad_perm<0x4E>(b1 ? xa : xb) + gxv;   // gate q, all four lanes
        const float sg2 = (q == 2) ? 2.0f : 1.0f;
        float a = sigmoidf_(sg2 * s);
        a = (q == 2) ? 2.0f * a - 1.0f : a;
        const float gi = quad_perm<0x00>(a), gf = quad_perm<0x55>(a), gg = quad_perm<0xAA>(a), go = quad_perm<0xFF>(a);
        c = gf * c + gi * gg;
        const float h = go * tanhf_(c);
        // all four lanes of the quad hold the same h: unconditional stores (same LDS word; hs keeps one word per lane) so
        // that no divergent branch hides the outstanding store from the compiler's vmcnt bookkeeping
        hbuf[(t + 1) & 1][j] = h;
        hs[(long long)t * 512 + tid] = h;
        gxv = gnext;
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait out the latency of the hs store and of the
        // gx prefetch on every one of the T dependent steps
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// output layer: p[t] = sigmoid(w_out . relu(h_t) + b); one wave per window
__global__ __launch_bounds__(256) void vad_out_kernel(const float* __restrict__ hs, const float* __restrict__ out_w, float out_b,
                                                       float* __restrict__ probs, int T) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float* h = hs + (long long)t * 512;                      // [unit j][4 identical copies]
    float v = out_w[lane] * fmaxf(h[4 * lane], 0.0f) + out_w[64 + lane] * fmaxf(h[4 * (64 + lane)], 0.0f);
    v = dpp_wave_sum(v);
    if (lane == 0) probs[t] = 1.0f / (1.0f + expf(-(v + out_b)));
}

}  // namespace wlx

// ---------------------------------------------------------------------------------------------------------------------
// C-ABI
using namespace wlx;

struct wlx_vad {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<void*> pool;
    VadW W{};
    float *d_pcm = nullptr, *d_gx = nullptr, *d_hs = nullptr, *d_probs = nullptr;
    float* h_pin = nullptr;              // pinned staging: PCM in, probabilities out
    long long cap_samples = 0;
    std::mutex mu;
};

#define VCK(call)                                                                                              \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return set_error(WLX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static int vad_upload(wlx_vad* v, const std::vector<float>& host, const float** out) {
    void* p = nullptr;
    VCK(hipMalloc(&p, host.size() * sizeof(float)));
    v->pool.push_back(p);
    VCK(hipMemcpyAsync(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, v->stream));   // never the legacy stream
    VCK(hipStreamSynchronize(v->stream));
    *out = reinterpret_cast<const float*>(p);
    return WLX_OK;
}

static int vad_reserve(wlx_vad* v, long long n_samples) {
    if (n_samples <= v->cap_samples) return WLX_OK;
    long long cap = 16000LL * 64;                                    // 64 s covers every streaming chunk (<= 45 s buffer)
    while (cap < n_samples) cap *= 2;
    for (float* p : {v->d_pcm, v->d_gx, v->d_hs, v->d_probs})
        if (p) (void)hipFree(p);
    if (v->h_pin) (void)hipHostFree(v->h_pin);
    v->d_pcm = v->d_gx = v->d_hs = v->d_probs = v->h_pin = nullptr;
    v->cap_samples = 0;
    const long long wins = cap / VAD_WINDOW + 1;
    VCK(hipMalloc((void**)&v->d_pcm, cap * sizeof(float)));
    VCK(hipMalloc((void**)&v->d_gx, wins * 512 * sizeof(float)));
    VCK(hipMalloc((void**)&v->d_hs, wins * 512 * sizeof(float)));
    VCK(hipMalloc((void**)&v->d_probs, wins * sizeof(float)));
    VCK(hipHostMalloc((void**)&v->h_pin, cap * sizeof(float), hipHostMallocDefault));
    v->cap_samples = cap;
    return WLX_OK;
}

extern "C" int32_t wlx_vad_create(const wlx_vad_weights* w, int32_t device, wlx_vad** out) {
    if (!w || !out) return set_error(WLX_ERR_ARG, "wlx_vad_create: null argument");
    const float* need[] = {w->stft_basis, w->enc_w[0], w->enc_w[1], w->enc_w[2], w->enc_w[3], w->enc_b[0], w->enc_b[1],
                           w->enc_b[2], w->enc_b[3], w->lstm_w_ih, w->lstm_w_hh, w->lstm_b_ih, w->lstm_b_hh, w->out_w, w->out_b};
    for (const float* p : need)
        if (!p) return set_error(WLX_ERR_WEIGHT, "wlx_vad_create: a weight pointer is null");
    VCK(hipSetDevice(device));
    wlx_vad* v = new wlx_vad();
    v->device = device;
    auto bail = [&](int rc) { wlx_vad_destroy(v); return rc; };
    if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&v->ev0) != hipSuccess ||
        hipEventCreate(&v->ev1) != hipSuccess)
        return bail(set_error(WLX_ERR_HIP, "wlx_vad_create: stream/event creation failed"));
    // ---- repack into the kernel layouts
    std::vector<float> t;
    int rc;
    t.assign(256 * VAD_BSTRIDE, 0.0f);
    for (int r = 0; r < 258; ++r)
        for (int n = 0; n < 256; ++n) t[n * VAD_BSTRIDE + r] = w->stft_basis[r * 256 + n];
    if ((rc = vad_upload(v, t, &v->W.basisT))) return bail(rc);
    auto conv_t = [&](const float* src, int cout, int cin, int j0, int nj, const float** dst) {   // [co][ci][3] -> [ci][nj][co]
        t.assign((size_t)cin * nj * cout, 0.0f);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int j = 0; j < nj; ++j) t[((size_t)ci * nj + j) * cout + co] = src[((size_t)co * cin + ci) * 3 + j0 + j];
        return vad_upload(v, t, dst);
    };
    if ((rc = conv_t(w->enc_w[0], 128, 129, 0, 3, &v->W.w1T))) return bail(rc);
    if ((rc = conv_t(w->enc_w[1], 64, 128, 0, 3, &v->W.w2T))) return bail(rc);
    if ((rc = conv_t(w->enc_w[2], 64, 64, 1, 2, &v->W.w3T))) return bail(rc);
    if ((rc = conv_t(w->enc_w[3], 128, 64, 1, 1, &v->W.w4T))) return bail(rc);
    const int bn[4] = {128, 64, 64, 128};
    const float** bd[4] = {&v->W.b1, &v->W.b2, &v->W.b3, &v->W.b4};
    for (int i = 0; i < 4; ++i) {
        t.assign(w->enc_b[i], w->enc_b[i] + bn[i]);
        if ((rc = vad_upload(v, t, bd[i]))) return bail(rc);
    }
    t.assign(128 * 512, 0.0f);
    for (int r = 0; r < 512; ++r)
        for (int k = 0; k < 128; ++k) t[k * 512 + r] = w->lstm_w_ih[r * 128 + k];
    if ((rc = vad_upload(v, t, &v->W.wihT))) return bail(rc);
    t.assign(512, 0.0f);
    for (int r = 0; r < 512; ++r) t[r] = w->lstm_b_ih[r] + w->lstm_b_hh[r];
    if ((rc = vad_upload(v, t, &v->W.bias))) return bail(rc);
    t.assign(4 * 32 * 512, 0.0f);
    for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 32; ++e)
            for (int lane = 0; lane < 512; ++lane) {
                const int j = lane >> 2, q = lane & 3;
                t[(g * 32 + e) * 512 + lane] = w->lstm_w_hh[(g * 128 + j) * 128 + 32 * q + e];
            }
    if ((rc = vad_upload(v, t, &v->W.whhP))) return bail(rc);
    t.assign(w->out_w, w->out_w + 128);
    if ((rc = vad_upload(v, t, &v->W.out_w))) return bail(rc);
    v->W.out_b = w->out_b[0];
    if ((rc = vad_reserve(v, 16000LL * 64))) return bail(rc);
    *out = v;
    return WLX_OK;
}

extern "C" void wlx_vad_destroy(wlx_vad* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    for (void* p : v->pool) (void)hipFree(p);
    for (float* p : {v->d_pcm, v->d_gx, v->d_hs, v->d_probs})
        if (p) (void)hipFree(p);
    if (v->h_pin) (void)hipHostFree(v->h_pin);
    if (v->ev0) (void)hipEventDestroy(v->ev0);
    if (v->ev1) (void)hipEventDestroy(v->ev1);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
}

extern "C" int32_t wlx_vad_probs(wlx_vad* v, const float* pcm, int64_t n, float* probs_out, int32_t cap,
                                 int32_t* n_windows_out, float* device_ms_out) {
    if (!v || !n_windows_out || (n > 0 && (!pcm || !probs_out))) return set_error(WLX_ERR_ARG, "wlx_vad_probs: null argument");
    if (n < 0) return set_error(WLX_ERR_ARG, "wlx_vad_probs: negative sample count");
    const long long T = (n + VAD_WINDOW - 1) / VAD_WINDOW;
    *n_windows_out = (int32_t)T;
    if (device_ms_out) *device_ms_out = 0.0f;
    if (T == 0) return WLX_OK;
    if (T > cap) return set_error(WLX_ERR_ARG, "wlx_vad_probs: %lld windows do not fit the output buffer (%d)", T, cap);
    std::lock_guard<std::mutex> lk(v->mu);
    VCK(hipSetDevice(v->device));
    (void)hipGetLastError();                  // a stale error of this thread must not be blamed on the launches below
    int rc = vad_reserve(v, n);
    if (rc) return rc;
    memcpy(v->h_pin, pcm, (size_t)n * sizeof(float));
    VCK(hipMemcpyAsync(v->d_pcm, v->h_pin, (size_t)n * sizeof(float), hipMemcpyHostToDevice, v->stream));
    VCK(hipEventRecord(v->ev0, v->stream));
    const int fe_blocks = (int)((T + VAD_WT - 1) / VAD_WT);
    hipLaunchKernelGGL(vad_frontend_kernel, dim3(fe_blocks), dim3(VAD_FE_THREADS), 0, v->stream, v->d_pcm, (long long)n, (int)T, v->W, v->d_gx);
    hipLaunchKernelGGL(vad_lstm_kernel, dim3(1), dim3(512), 0, v->stream, v->d_gx, v->W.whhP, v->d_hs, (int)T);
    hipLaunchKernelGGL(vad_out_kernel, dim3((int)((T + 3) / 4)), dim3(256), 0, v->stream, v->d_hs, v->W.out_w, v->W.out_b, v->d_probs, (int)T);
    VCK(hipGetLastError());
    VCK(hipEventRecord(v->ev1, v->stream));
    VCK(hipMemcpyAsync(v->h_pin, v->d_probs, (size_t)T * sizeof(float), hipMemcpyDeviceToHost, v->stream));
    VCK(hipStreamSynchronize(v->stream));
    memcpy(probs_out, v->h_pin, (size_t)T * sizeof(float));
    if (device_ms_out) VCK(hipEventElapsedTime(device_ms_out, v->ev0, v->ev1));
    return WLX_OK;
}

// The same on samples that are already in HBM: [start, start + n) of a client's PCM ring (engine.hip wlx_ring_*, include/wlx.h) — the
// gate of a streaming session reads the ring the socket thread filled, no second host-to-device copy.
extern "C" int32_t wlx_vad_probs_resident(wlx_vad* v, wlx_ring* r, int64_t start, int64_t n, int32_t extra_zero_windows, float* probs_out, int32_t cap,
                                          int32_t* n_windows_out, float* device_ms_out) {
    if (!v || !r || !n_windows_out || (n > 0 && !probs_out)) return set_error(WLX_ERR_ARG, "wlx_vad_probs_resident: null argument");
    if (n < 0) return set_error(WLX_ERR_ARG, "wlx_vad_probs_resident: negative sample count");
    if (r->device != v->device) return set_error(WLX_ERR_ARG, "wlx_vad_probs_resident: the ring lives on device %d, the VAD model on %d", r->device, v->device);
    if (extra_zero_windows < 0 || extra_zero_windows > 4) return set_error(WLX_ERR_ARG, "wlx_vad_probs_resident: extra_zero_windows out of range");
    const long long T = (n + VAD_WINDOW - 1) / VAD_WINDOW + extra_zero_windows;
    *n_windows_out = (int32_t)T;
    if (device_ms_out) *device_ms_out = 0.0f;
    if (T == 0) return WLX_OK;
    if (T > cap) return set_error(WLX_ERR_ARG, "wlx_vad_probs_resident: %lld windows do not fit the output buffer (%d)", T, cap);
    std::lock_guard<std::mutex> lk(v->mu);
    std::lock_guard<std::mutex> lr(r->mu);          // no append / trim while the kernels read (the call waits for them below)
    if (start < r->base || start + n > r->base + r->resident)
        return set_error(WLX_ERR_STATE, "wlx_vad_probs_resident: [%lld, %lld) is not resident (ring holds [%lld, %lld))", (long long)start,
                         (long long)(start + n), (long long)r->base, (long long)(r->base + r->resident));
    VCK(hipSetDevice(v->device));
    (void)hipGetLastError();
    int rc = vad_reserve(v, n + (long long)extra_zero_windows * VAD_WINDOW);
    if (rc) return rc;
    const float* pcm = r->buf + (start - r->base);
    VCK(hipEventRecord(v->ev0, v->stream));
    const int fe_blocks = (int)((T + VAD_WT - 1) / VAD_WT);
    hipLaunchKernelGGL(vad_frontend_kernel, dim3(fe_blocks), dim3(VAD_FE_THREADS), 0, v->stream, pcm, (long long)n, (int)T, v->W, v->d_gx);
    hipLaunchKernelGGL(vad_lstm_kernel, dim3(1), dim3(512), 0, v->stream, v->d_gx, v->W.whhP, v->d_hs, (int)T);
    hipLaunchKernelGGL(vad_out_kernel, dim3((int)((T + 3) / 4)), dim3(256), 0, v->stream, v->d_hs, v->W.out_w, v->W.out_b, v->d_probs, (int)T);
    VCK(hipGetLastError());
    VCK(hipEventRecord(v->ev1, v->stream));
    VCK(hipMemcpyAsync(v->h_pin, v->d_probs, (size_t)T * sizeof(float), hipMemcpyDeviceToHost, v->stream));
    VCK(hipStreamSynchronize(v->stream));
    memcpy(probs_out, v->h_pin, (size_t)T * sizeof(float));
    if (device_ms_out) VCK(hipEventElapsedTime(device_ms_out, v->ev0, v->ev1));
    return WLX_OK;
}

// ------------------------------------------------------------------ host: hysteresis segmentation of the probabilities
// whisperlive_amd/vad.py speech_segments_from_probs (= faster_whisper.vad.get_speech_timestamps' loop, as the reference calls it at
// whisper_live/transcriber/transcriber_faster_whisper.py:825-852) statement for statement, in C: the loop runs on the host BETWEEN the VAD launch
// and the log-mel launch — the GPU waits for it — and costs 36 us for the 250 windows of an 8 s chunk and 136 us for 30 s in Python, ~2 us here
// (round 6). The thresholds arrive as doubles already rounded to float32 (numpy compares a float32 probability with a Python float in float32);
// the durations are the Python code's floats (max_speech may be +inf); positions are integers. Python's `gap // 2` floors.
extern "C" int32_t wlx_vad_segments(const float* probs, int32_t n_windows, int64_t n_samples, double thr, double neg, double min_speech, double pad,
                                    double max_speech, double min_silence, double min_silence_at_max, int64_t* start_end_out, int32_t cap,
                                    int32_t* n_out) {
    if (!probs || n_windows < 0 || !start_end_out || !n_out || cap < 0) return set_error(WLX_ERR_ARG, "wlx_vad_segments: bad argument");
    const int64_t W = 512;
    int32_t ns = 0;
    auto push = [&](int64_t a, int64_t b) -> bool { if (ns >= cap) return false; start_end_out[2 * ns] = a; start_end_out[2 * ns + 1] = b; ++ns; return true; };
    bool have_cur = false, active = false;
    int64_t cur_start = 0, silence_from = 0, cut_at = 0, resume_at = 0;
    for (int32_t i = 0; i < n_windows; ++i) {
        const double p = (double)probs[i];
        const int64_t pos = W * i;
        if (p >= thr && silence_from) {
            silence_from = 0;
            if (resume_at < cut_at) resume_at = pos;
        }
        if (p >= thr && !active) { active = true; cur_start = pos; have_cur = true; continue; }
        if (active && (double)(pos - cur_start) > max_speech) {
            if (cut_at) {
                if (!push(cur_start, cut_at)) return set_error(WLX_ERR_ARG, "wlx_vad_segments: more segments than the output holds");
                have_cur = false;
                if (resume_at < cut_at) active = false;
                else { cur_start = resume_at; have_cur = true; }
                cut_at = resume_at = silence_from = 0;
            } else {
                if (!push(cur_start, pos)) return set_error(WLX_ERR_ARG, "wlx_vad_segments: more segments than the output holds");
                have_cur = false;
                cut_at = resume_at = silence_from = 0;
                active = false;
                continue;
            }
        }
        if (p < neg && active) {
            if (!silence_from) silence_from = pos;
            if ((double)(pos - silence_from) > min_silence_at_max) cut_at = silence_from;
            if ((double)(pos - silence_from) < min_silence) continue;
            if ((double)(silence_from - cur_start) > min_speech) { if (!push(cur_start, silence_from)) return set_error(WLX_ERR_ARG, "wlx_vad_segments: more segments than the output holds"); }
            have_cur = false;
            cut_at = resume_at = silence_from = 0;
            active = false;
        }
    }
    if (have_cur && (double)(n_samples - cur_start) > min_speech) { if (!push(cur_start, n_samples)) return set_error(WLX_ERR_ARG, "wlx_vad_segments: more segments than the output holds"); }
    auto floordiv2 = [](int64_t g) -> int64_t { return (g >= 0) ? g / 2 : -((-g + 1) / 2); };
    auto clamp0 = [](double v) -> int64_t { return (int64_t)(v > 0.0 ? v : 0.0); };
    for (int32_t i = 0; i < ns; ++i) {
        int64_t& s_ = start_end_out[2 * i]; int64_t& e_ = start_end_out[2 * i + 1];
        if (i == 0) s_ = clamp0((double)s_ - pad);
        if (i != ns - 1) {
            int64_t& s2 = start_end_out[2 * i + 2];
            const int64_t gap = s2 - e_;
            if ((double)gap < 2.0 * pad) {
                e_ += floordiv2(gap);
                const int64_t t = s2 - floordiv2(gap);
                s2 = t > 0 ? t : 0;
            } else {
                const double ee = (double)e_ + pad;
                e_ = (int64_t)(ee < (double)n_samples ? ee : (double)n_samples);
                s2 = clamp0((double)s2 - pad);
            }
        } else {
            const double ee = (double)e_ + pad;
            e_ = (int64_t)(ee < (double)n_samples ? ee : (double)n_samples);
        }
    }
    *n_out = ns;
    return WLX_OK;
}
