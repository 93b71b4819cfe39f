// gemm.hip — fp16 MFMA GEMM for the encoder (ctranslate2 Whisper.encode replacement,
// whisper_live/transcriber/transcriber_faster_whisper.py:1339-1348) plus row LayerNorm.
//
// D[n][m] = sum_k W[n][k] * X[m][k]  ("swapped" operands, see common.h): weights are read as
// pre-packed 1 KiB fragments straight into registers, activations as 16-byte row segments of a
// row-major fp16 matrix (lda is free, which is how the two conv1d layers run without im2col:
// a conv window of 3 consecutive time steps of a time-major activation IS one contiguous K-row
// of length 3*C starting at row t*stride — the GEMM just uses lda = stride*C).
// Both operands are staged through LDS in MFMA fragment order, two 32-wide k-tiles per stage, double buffered through
// registers: the next stage's global loads are in flight under the current stage's MFMAs, one barrier per stage.
// Measured alternatives on the Whisper-small encoder (MI355X, average per launch): every fragment straight from L2/L1
// per wave (first form; 128 B/clk/CU demanded of a 64 B/clk L1) 39 us; THIS form 32 us; a 4-deep LDS ring filled by
// LDS-direct loads (global_load_lds_dwordx4) 34-38 us (LDS-direct fills are issue-limited per wave); two register sets
// (two stages of loads in flight) 36-48 us (188 VGPRs: occupancy). At M = 1500 and N = 768..3072 the launches are
// short (12-96 stages per workgroup) and still latency-dominated. Forcing one tile shape for every GEMM of the encoder
// (128x128 / 128x64 / 64x64 workgroup tiles) moves the encoder by 2 % at most (2.21 ms chosen per shape as below, 2.19 / 2.17 ms
// all-128x64 / all-64x64, 2.62 ms all-128x128): the stage loop's latency, not the tiling or the 1.1-wave grids, bounds it.
// Wave tile = WNT x WMT 16x16 tiles; a 4-wave workgroup covers (2*WNT*16) x (2*WMT*16) outputs. Epilogues are fused
// (bias, exact GELU,
// positional add, residual accumulate in fp32, q-scaling, K/V scatter with V stored transposed
// for the attention kernels).
#include "kernels.h"

namespace wlx {

template <int WNT, int WMT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    constexpr int KS = 2;                       // k-tiles (of 32) per LDS stage
    constexpr int FA = 2 * WNT * KS;            // weight fragments per stage (1 KiB each)
    constexpr int FB = 2 * WMT * KS;            // activation fragments per stage
    constexpr int CA = FA / 4, CB = FB / 4;     // fragments each of the 4 waves copies per stage
    extern __shared__ __attribute__((aligned(16))) f16x8 stage_lds[];   // [2][FA + FB][64 lanes] x 16 B
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int wn = wave & 1, wm = wave >> 1;
    const int NT_total = (p.N + 15) >> 4;
    const int ntb = blockIdx.x * (2 * WNT);     // first n-tile of the workgroup
    const int mb = blockIdx.y * (2 * WMT * 16); // first row of the workgroup
    const int nt0 = ntb + wn * WNT;
    const int m0 = mb + wm * (WMT * 16);
    const int z = blockIdx.z;
    const half_t* A = p.A + (long)z * p.strideA;
    const int KT = p.KT;

    // global -> register -> LDS staging: wave w copies fragments w, w + 4, ... of the stage; both operands land in LDS
    // in MFMA fragment order (weights are stored that way; an activation fragment is 16 rows x 64 contiguous bytes),
    // so every LDS access of the kernel is a linear, conflict-free 16 bytes per lane.
    const half_t* asrc[CA];
    const half_t* bsrc[CB];
#pragma unroll
    for (int j = 0; j < CA; ++j) {
        const int f = wave + 4 * j, ni = f / KS, kk = f % KS;
        int nt = ntb + ni;
        if (nt >= NT_total) nt = NT_total - 1;
        asrc[j] = p.Wp + ((long)nt * KT + kk) * 512 + lane * 8;
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int f = wave + 4 * j, mi = f / KS, kk = f % KS;
        int row = mb + mi * 16 + c;
        if (row >= p.M) row = p.M - 1;
        bsrc[j] = A + (long)row * p.lda + kk * 32 + g * 8;
    }
    f16x8 ra[CA], rb[CB];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto gload = [&](int kt0) {
#pragma unroll
        for (int j = 0; j < CA; ++j) {
            const int kk = (wave + 4 * j) % KS;
            ra[j] = (kt0 + kk < KT) ? ld_f16x8(asrc[j] + (long)kt0 * 512) : zero8;
        }
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int kk = (wave + 4 * j) % KS;
            rb[j] = (kt0 + kk < KT) ? ld_f16x8(bsrc[j] + (long)kt0 * 32) : zero8;
        }
    };
    auto lstore = [&](int buf) {
        f16x8* dst = stage_lds + (long)buf * (FA + FB) * 64 + lane;
#pragma unroll
        for (int j = 0; j < CA; ++j) dst[(wave + 4 * j) * 64] = ra[j];
#pragma unroll
        for (int j = 0; j < CB; ++j) dst[(FA + wave + 4 * j) * 64] = rb[j];
    };

    f32x4 acc[WNT][WMT];
#pragma unroll
    for (int ni = 0; ni < WNT; ++ni)
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int S = (KT + KS - 1) / KS;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int st = 0; st < S; ++st) {
        const bool more = st + 1 < S;
#ifndef WLX_PROBE_NO_GLOAD   // (WLX_PROBE_*: scripts/ubench/gemm_probe.hip removes one pipeline phase at a time; never defined in libwlx)
        if (more) gload((st + 1) * KS);                                   // next stage in flight under this stage's MFMAs
#endif
        const f16x8* src = stage_lds + (long)(st & 1) * (FA + FB) * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            f16x8 wf[WNT], af[WMT];
#pragma unroll
#ifndef WLX_PROBE_NO_LREAD
            for (int ni = 0; ni < WNT; ++ni) wf[ni] = src[((wn * WNT + ni) * KS + kk) * 64];
#pragma unroll
            for (int mi = 0; mi < WMT; ++mi) af[mi] = src[(FA + (wm * WMT + mi) * KS + kk) * 64];
#else
            for (int ni = 0; ni < WNT; ++ni) wf[ni] = ra[ni % CA];
#pragma unroll
            for (int mi = 0; mi < WMT; ++mi) af[mi] = rb[mi % CB];
#endif
#pragma unroll
            for (int ni = 0; ni < WNT; ++ni)
#pragma unroll
                for (int mi = 0; mi < WMT; ++mi) {
#ifndef WLX_PROBE_NO_MFMA
                    acc[ni][mi] = mfma16(wf[ni], af[mi], acc[ni][mi]);
#else
                    asm volatile("" :: "v"(wf[ni]), "v"(af[mi]));
#endif
                }
        }
#ifndef WLX_PROBE_NO_LSTORE
        if (more) lstore((st + 1) & 1);                                   // the other buffer: last read one barrier ago
#endif
#ifndef WLX_PROBE_NO_BARRIER
        __syncthreads();
#endif
    }
#ifdef WLX_PROBE_NO_EPILOGUE
    {
        float t = 0.f;
#pragma unroll
        for (int ni = 0; ni < WNT; ++ni)
#pragma unroll
            for (int mi = 0; mi < WMT; ++mi) t += acc[ni][mi][0] + acc[ni][mi][1] + acc[ni][mi][2] + acc[ni][mi][3];
        if (t == 1.2345e30f) p.X[0] = t;
        return;
    }
#endif

    // ---------------- epilogue: lane owns columns n..n+3 of row m
#pragma unroll
    for (int ni = 0; ni < WNT; ++ni) {
        const int n = (nt0 + ni) * 16 + g * 4;
        if (n >= p.N) continue;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w;
        }
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi) {
            const int m = m0 + mi * 16 + c;
            if (m >= p.M) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ni][mi][r] + b4[r];
            switch (p.mode) {
                case GEMM_STORE_F16:
                case GEMM_GELU_F16: {
                    if (p.mode == GEMM_GELU_F16) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                    }
                    f16x4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    *reinterpret_cast<f16x4*>(p.C + (long)z * p.strideC + (long)m * p.ldc + n) = o;
                } break;
                case GEMM_GELU_POS_F32: {
                    float4 pv = *reinterpret_cast<const float4*>(p.pos + (long)m * p.N + n);
                    float4 o = make_float4(gelu_erf(v[0]) + pv.x, gelu_erf(v[1]) + pv.y,
                                           gelu_erf(v[2]) + pv.z, gelu_erf(v[3]) + pv.w);
                    *reinterpret_cast<float4*>(p.X + (long)z * p.strideX + (long)m * p.ldx + n) = o;
                } break;
                case GEMM_RESID_F32: {
                    float4* xp = reinterpret_cast<float4*>(p.X + (long)z * p.strideX + (long)m * p.ldx + n);
                    float4 o = *xp;
                    o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                    *xp = o;
                } break;
                case GEMM_QKV: {
                    const int item = m / p.rows_per_item, t = m - item * p.rows_per_item;
                    if (n < p.d) {
                        f16x4 o = {(half_t)(v[0] * p.qscale), (half_t)(v[1] * p.qscale),
                                   (half_t)(v[2] * p.qscale), (half_t)(v[3] * p.qscale)};
                        *reinterpret_cast<f16x4*>(p.C + (long)m * p.ldc + n) = o;
                    } else if (n < 2 * p.d) {
                        f16x4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        *reinterpret_cast<f16x4*>(p.Kout + (long)item * p.kv_item_stride_k + (long)t * p.ldk + (n - p.d)) = o;
                    } else {
                        half_t* vt = p.Vt + (long)item * p.kv_item_stride_v + (long)(n - 2 * p.d) * p.ldvt + t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) vt[(long)r * p.ldvt] = (half_t)v[r];
                    }
                } break;
                case GEMM_CROSS_KV: {
                    // TILE-PACKED cross K / V for the decode cross-attention (decoder.hip dec_cross_attn_kernel): per
                    // (layer, item, head, 32-key tile) a 4 KiB image in MFMA operand order —
                    //   K: [s2][kt2][lane = g*16 + c][e]  = K[key = tile*32 + s2*16 + c][dim = kt2*32 + g*8 + e]
                    //   V: [dt][lane = g*16 + c][e]       = V[key = tile*32 + (e < 4 ? g*4 + e : 16 + g*4 + e - 4)][dim = dt*16 + c]
                    const int item = m / p.rows_per_item, t = m - item * p.rows_per_item;
                    const int l = n / (2 * p.d), nn = n - l * 2 * p.d;
                    const bool isk = nn < p.d;
                    const int da = isk ? nn : nn - p.d;
                    const int hd = da >> 6, dd = da & 63;
                    const int tile = t >> 5, k32 = t & 31;
                    const long tbase = ((long)hd * (WLX_T_AUDIO_PAD / 32) + tile) * 2048;
                    if (isk) {
                        const int s2 = k32 >> 4, cc = k32 & 15, kt2 = dd >> 5, gg = (dd & 31) >> 3, e0 = dd & 7;
                        f16x4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        *reinterpret_cast<f16x4*>(p.Kout + (long)l * p.kv_layer_stride_k + (long)item * p.kv_item_stride_k + tbase +
                                                  ((s2 * 2 + kt2) * 64 + gg * 16 + cc) * 8 + e0) = o;
                    } else {
                        const int dt = dd >> 4, c0 = dd & 15;
                        const int gg = (k32 & 15) >> 2, ee = (k32 & 3) + ((k32 >> 4) << 2);
                        half_t* vp = p.Vt + (long)l * p.kv_layer_stride_v + (long)item * p.kv_item_stride_v + tbase +
                                     (dt * 64 + gg * 16 + c0) * 8 + ee;
#pragma unroll
                        for (int r = 0; r < 4; ++r) vp[r * 8] = (half_t)v[r];
                    }
                } break;
                default: break;
            }
        }
    }
}

void launch_gemm(const GemmParams& p, int zbatch, hipStream_t s) {
    const int NT_total = (p.N + 15) / 16;
    // tile choice: wide outputs get 128x128 workgroup tiles (4x4 wave tiles: 16 MFMAs per 8 fragment
    // loads); narrow ones (N = d) use 64x64 so the launch still spreads over the 256 CUs.
    long blocks_big = (long)((NT_total + 7) / 8) * ((p.M + 127) / 128) * zbatch;
    if (blocks_big >= 200) {
        dim3 grid((NT_total + 7) / 8, (p.M + 127) / 128, zbatch);
        hipLaunchKernelGGL((gemm_kernel<4, 4>), grid, dim3(256), 2 * (16 + 16) * 1024, s, p);
    } else {
        long blocks_mid = (long)((NT_total + 7) / 8) * ((p.M + 63) / 64) * zbatch;
        if (blocks_mid >= 200) {
            dim3 grid((NT_total + 7) / 8, (p.M + 63) / 64, zbatch);
            hipLaunchKernelGGL((gemm_kernel<4, 2>), grid, dim3(256), 2 * (16 + 8) * 1024, s, p);
        } else {
            dim3 grid((NT_total + 3) / 4, (p.M + 63) / 64, zbatch);
            hipLaunchKernelGGL((gemm_kernel<2, 2>), grid, dim3(256), 2 * (8 + 8) * 1024, s, p);
        }
    }
}

// ---------------------------------------------------------------- LayerNorm (wave per row)
template <bool OUT32>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        half_t* __restrict__ out16, float* __restrict__ out32,
                                                        long ldo, int M, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * ldx);
    const int n4 = d >> 2;
    float4 v[8];  // d <= 2048
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int j = lane + i * 64;
        if (j < n4) { v[i] = xr[j]; s += v[i].x + v[i].y + v[i].z + v[i].w; }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int j = lane + i * 64;
        if (j < n4) {
            float a = v[i].x - mean, b = v[i].y - mean, c2 = v[i].z - mean, e = v[i].w - mean;
            q += a * a + b * b + c2 * c2 + e * e;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int j = lane + i * 64;
        if (j < n4) {
            float4 gg = g4[j], bb = b4[j];
            float o0 = (v[i].x - mean) * rstd * gg.x + bb.x;
            float o1 = (v[i].y - mean) * rstd * gg.y + bb.y;
            float o2 = (v[i].z - mean) * rstd * gg.z + bb.z;
            float o3 = (v[i].w - mean) * rstd * gg.w + bb.w;
            f16x4 o = {(half_t)o0, (half_t)o1, (half_t)o2, (half_t)o3};
            *reinterpret_cast<f16x4*>(out16 + (long)row * ldo + j * 4) = o;
            if (OUT32) *reinterpret_cast<float4*>(out32 + (long)row * ldo + j * 4) = make_float4(o0, o1, o2, o3);
        }
    }
}

void launch_layernorm_f16(const float* x, long ldx, const float* gamma, const float* beta,
                          half_t* out, long ldo, int M, int d, hipStream_t s) {
    hipLaunchKernelGGL((layernorm_kernel<false>), dim3((M + 3) / 4), dim3(256), 0, s, x, ldx, gamma, beta, out,
                       (float*)nullptr, ldo, M, d);
}
void launch_layernorm_f16_f32(const float* x, long ldx, const float* gamma, const float* beta,
                              half_t* out16, float* out32, long ldo, int M, int d, hipStream_t s) {
    hipLaunchKernelGGL((layernorm_kernel<true>), dim3((M + 3) / 4), dim3(256), 0, s, x, ldx, gamma, beta, out16,
                       out32, ldo, M, d);
}

}  // namespace wlx
