// logmel.hip — Whisper log-mel front-end on gfx950.
//
// Replaces faster_whisper.feature_extractor.FeatureExtractor.__call__(waveform, padding=160)
// as called at whisper_live/transcriber/transcriber_faster_whisper.py:862 (also :426, :1759;
// whisper_live/batch_inference.py:258). Recipe (in-tree restatement without the 160 pad:
// whisper_live/transcriber/tensorrt_utils.py:177-190):
//   x = pad(x, (0,160)); STFT(n_fft=400, hop=160, periodic Hann, center/reflect 200); drop last
//   frame; |.|^2; slaney-mel @ power; log10(max(.,1e-10)); max(., global_max-8); (.+4)/4.
//
// Kernel structure: one workgroup per tile of FT=8 frames. The frames overlap 240/400 samples,
// so the tile's 1520 PCM samples are loaded once, coalesced, from HBM into LDS, windowed there,
// and every thread owns one frequency bin for all 8 frames (400-point real DFT as a table-driven
// dot product: the 8 frames share each twiddle read, the frame samples are LDS broadcasts).
// The mel reduction is sparse (only the non-zero span of each triangular filter is visited).
// The clamp needs the max over the WHOLE chunk, so a per-workgroup max is folded into one
// global word with an order-preserving atomic and applied by a second, elementwise kernel.
// The path is HBM-trivial (1.9 MB in, 0.96 MB out per 30 s) and launch-latency bound.
#include "kernels.h"

namespace wlx {

#define LM_FT 8
#define LM_NFFT 400
#define LM_HOP 160
#define LM_NBINS 201
#define LM_PW_LD 208

__device__ __forceinline__ long reflect_idx(long j, long L) {
    // numpy "reflect" padding, valid for any pad width (periodic reflection without edge repeat)
    if (L <= 1) return 0;
    long p = 2 * (L - 1);
    j %= p;
    if (j < 0) j += p;
    if (j >= L) j = p - j;
    return j;
}

__device__ __forceinline__ void atomic_max_float(unsigned* addr, float v) {
    if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(addr, __float_as_uint(v));
}

// (round 4) the kernels take a small by-value table of items, blockIdx.y = item: the 12 windows of a batched encode are ONE launch of each
// kernel instead of 12 (x 3 kernels), and their tiles run side by side (a tile is latency-bound: 58 us whether 376 or 4512 of them are resident)
__global__ __launch_bounds__(256) void logmel_kernel(LogmelBatch lb, int n_mels,
                                                     const float* __restrict__ window,
                                                     const float* __restrict__ twiddle,
                                                     const float* __restrict__ filters,
                                                     const int* __restrict__ frange, long ld) {
    const int item = blockIdx.y;
    const float* __restrict__ pcm = lb.pcm[item];
    const long n = lb.n[item];
    float* __restrict__ feats = lb.feats[item];
    const int T = lb.T[item];
    unsigned* __restrict__ gmax = lb.gmax[item];
    const long long* __restrict__ rng = lb.rng[item];
    const int nr = lb.nr[item];
    if ((int)blockIdx.x * LM_FT >= T) return;
    __shared__ __attribute__((aligned(16))) float xw[LM_FT][LM_NFFT];
    __shared__ __attribute__((aligned(16))) float2 tw[LM_NFFT];
    __shared__ float pw[LM_FT][LM_PW_LD];

    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * LM_FT;
    const long L = n + 160;  // padded length

    for (int i = tid; i < LM_NFFT; i += 256) tw[i] = make_float2(twiddle[2 * i], twiddle[2 * i + 1]);
    // windowed frames: frame f covers padded-signal samples [ (t0+f)*160 - 200, +400 )
    for (int i = tid; i < LM_FT * LM_NFFT; i += 256) {
        int f = i / LM_NFFT, j = i - f * LM_NFFT;
        long src = reflect_idx((long)(t0 + f) * LM_HOP + j - 200, L);
        float v = 0.0f;
        if (src < n) {
            long phys = src;
            if (nr > 0) {              // the PCM ring: sample `src` of the concatenated speech ranges lives at rng[2 q] + (src - rng[2 q + 1])
                int q = 0;
                for (int i = 1; i < nr; ++i) q = (src >= (long)rng[2 * i + 1]) ? i : q;
                phys = (long)rng[2 * q] + (src - (long)rng[2 * q + 1]);
            }
            v = pcm[phys];
        }
        xw[f][j] = v * window[j];
    }
    __syncthreads();

    const int k = tid;  // frequency bin
    if (k < LM_NBINS) {
        float re[LM_FT], im[LM_FT];
#pragma unroll
        for (int f = 0; f < LM_FT; ++f) { re[f] = 0.f; im[f] = 0.f; }
        int idx = 0;  // (k * j) mod 400
        for (int j4 = 0; j4 < LM_NFFT / 4; ++j4) {
            float2 c0 = tw[idx]; idx += k; if (idx >= LM_NFFT) idx -= LM_NFFT;
            float2 c1 = tw[idx]; idx += k; if (idx >= LM_NFFT) idx -= LM_NFFT;
            float2 c2 = tw[idx]; idx += k; if (idx >= LM_NFFT) idx -= LM_NFFT;
            float2 c3 = tw[idx]; idx += k; if (idx >= LM_NFFT) idx -= LM_NFFT;
#pragma unroll
            for (int f = 0; f < LM_FT; ++f) {
                float4 x = *reinterpret_cast<const float4*>(&xw[f][j4 * 4]);
                re[f] = fmaf(x.x, c0.x, re[f]); im[f] = fmaf(x.x, c0.y, im[f]);
                re[f] = fmaf(x.y, c1.x, re[f]); im[f] = fmaf(x.y, c1.y, im[f]);
                re[f] = fmaf(x.z, c2.x, re[f]); im[f] = fmaf(x.z, c2.y, im[f]);
                re[f] = fmaf(x.w, c3.x, re[f]); im[f] = fmaf(x.w, c3.y, im[f]);
            }
        }
#pragma unroll
        for (int f = 0; f < LM_FT; ++f) pw[f][k] = re[f] * re[f] + im[f] * im[f];
    }
    __syncthreads();

    float lmax = WLX_NEG_INF;
    for (int task = tid; task < n_mels * LM_FT; task += 256) {
        int m = task / LM_FT, f = task - m * LM_FT;
        int t = t0 + f;
        if (t >= T) continue;
        int lo = frange[2 * m], hi = frange[2 * m + 1];
        const float* fr = filters + (long)m * LM_NBINS;
        float acc = 0.f;
        for (int b = lo; b < hi; ++b) acc = fmaf(fr[b], pw[f][b], acc);
        float v = log10f(fmaxf(acc, 1e-10f));
        feats[(long)m * ld + t] = v;
        lmax = fmaxf(lmax, v);
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0 && lmax > WLX_NEG_INF) atomic_max_float(gmax, lmax);
}

__global__ void logmel_finalize_kernel(LogmelBatch lb, long ld, int n_mels) {
    const int item = blockIdx.y;
    float* __restrict__ feats = lb.feats[item];
    const int T = lb.T[item];
    const float floor_v = __uint_as_float(*lb.gmax[item]) - 8.0f;
    long total = (long)n_mels * T;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int m = i / T, t = i - (long)m * T;
        float v = feats[(long)m * ld + t];
        v = fmaxf(v, floor_v);
        feats[(long)m * ld + t] = (v + 4.0f) / 4.0f;
    }
}

__global__ void set_u32_kernel(LogmelBatch lb, unsigned v) { if ((int)threadIdx.x < lb.n_items) *lb.gmax[threadIdx.x] = v; }

void launch_logmel_batch(const LogmelBatch& lb, int n_mels, const LogmelConsts& c, long ld, hipStream_t s) {
    if (lb.n_items < 1) return;
    hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(64), 0, s, lb, 0xff800000u);  // -inf
    int Tmax = 0;
    for (int i = 0; i < lb.n_items; ++i) Tmax = lb.T[i] > Tmax ? lb.T[i] : Tmax;
    const int blocks = (Tmax + LM_FT - 1) / LM_FT;
    hipLaunchKernelGGL(logmel_kernel, dim3(blocks, lb.n_items), dim3(256), 0, s, lb, n_mels, c.window, c.twiddle, c.filters, c.frange, ld);
    const long total = (long)n_mels * Tmax;
    int fb = (int)((total + 255) / 256);
    if (fb > 1024) fb = 1024;
    hipLaunchKernelGGL(logmel_finalize_kernel, dim3(fb, lb.n_items), dim3(256), 0, s, lb, ld, n_mels);
}
void launch_logmel(const float* pcm, long n, int n_mels, const LogmelConsts& c, float* feats, long ld,
                   int T, unsigned* gmax, hipStream_t s) {
    LogmelBatch lb{};
    lb.n_items = 1; lb.pcm[0] = pcm; lb.n[0] = n; lb.feats[0] = feats; lb.T[0] = T; lb.gmax[0] = gmax;
    launch_logmel_batch(lb, n_mels, c, ld, s);
}

// feats[m][seek + t] (t < seg) -> featT[(1 + t) * n_mels + m] as fp16, zeros for seg <= t < 3000.
// Row 0 and row 3001 of featT are the conv1 zero padding and are never written here.
// (round 4) one launch for the items of a batched encode: blockIdx.y = item, the items' feature matrices / transposed windows at regular
// strides, the windows' (seek, seg) in a by-value table (12 launches of 10 us in front of a 12-window encoder -> one)
__global__ __launch_bounds__(256) void prep_window_kernel(const float* __restrict__ feats0, long ld, int n_mels, long item_stride,
                                                          PrepWindows w, half_t* __restrict__ featT0, long featT_stride) {
    // 32 frames per block (round 6; 64 before: 47 workgroups per window, a 10.7 us launch that is one latency chain long — 94 halve its trips)
    __shared__ float tile[128][33];
    const int item = blockIdx.y;
    const float* __restrict__ feats = feats0 + (long)item * item_stride;
    half_t* __restrict__ featT = featT0 + (long)item * featT_stride;
    const int seek = w.seek[item], seg = w.seg[item];
    const int tb = blockIdx.x * 32;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_mels * 32; i += 256) {
        int m = i >> 5, tl = i & 31;
        int t = tb + tl;
        tile[m][tl] = (t < seg) ? feats[(long)m * ld + seek + t] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < n_mels * 32; i += 256) {
        int tl = i / n_mels, m = i - tl * n_mels;
        int t = tb + tl;
        if (t < WLX_N_FRAMES) featT[(long)(1 + t) * n_mels + m] = (half_t)tile[m][tl];
    }
}

void launch_prep_windows(const float* feats0, long ld, int n_mels, long item_stride, const PrepWindows& w, int items,
                         half_t* featT0, long featT_stride, hipStream_t s) {
    int blocks = (WLX_N_FRAMES + 31) / 32;
    hipLaunchKernelGGL(prep_window_kernel, dim3(blocks, items), dim3(256), 0, s, feats0, ld, n_mels, item_stride, w, featT0, featT_stride);
}

}  // namespace wlx
