// decoder.hip — one autoregressive decoder step on gfx950 (the inner loop of
// ctranslate2.models.Whisper.generate, called from
// whisper_live/transcriber/transcriber_faster_whisper.py:1394-1407 and
// whisper_live/batch_inference.py:355-357; network: HF modeling_whisper.py:416-498).
//
// The step is HBM/latency bound: M = beams x items (5..40) rows against ~278 MB of fp16 weights
// (Whisper-small). Every projection is a skinny GEMM  Y[M x N] = X[M x K] * W^T  done with MFMA
// 16x16x32 in the swapped form (weights = A operand, read as pre-packed contiguous 1 KiB
// fragments straight into registers; the <=64 activation rows = B operand held in registers),
// K split across the waves of a workgroup and combined through LDS, so each weight byte is read
// exactly once per step and stores are deterministic (no atomics). LayerNorm is fused into the
// prologue of the consuming projection (the weight fragment loads are issued BEFORE the
// prologue so their HBM latency overlaps the statistics), bias / GELU / residual-accumulate /
// q-scaling / KV-cache append into the epilogue. Beam reordering never moves the KV cache: an
// int16 ancestry table maps (row, position) -> cache row. All per-step scalars (position,
// tokens, ancestry, done flag) live in device memory so one captured hipGraph replays every step.
#include "decoder.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <atomic>

namespace wlx {

// ------------------------------------------------------------------ embedding
__global__ __launch_bounds__(256) void dec_embed_kernel(const half_t* __restrict__ tok_emb,
                                                        const float* __restrict__ pos_emb, int d,
                                                        const int* __restrict__ token,
                                                        const int* __restrict__ pos,
                                                        const int* __restrict__ cache, int* __restrict__ intok,
                                                        float* __restrict__ x, const int* __restrict__ done WLX_TR_PARAM) {
    if (done && *done) return;
    WLX_TR_BEGIN();
    const int r = blockIdx.x;
    const int tok = token[r], p = pos[r];
    if (threadIdx.x == 0) intok[(long)cache[r] * WLX_T_TEXT + p] = tok;
    const half_t* te = tok_emb + (long)tok * d;
    const float* pe = pos_emb + (long)p * d;
    for (int i = threadIdx.x * 4; i < d; i += 256 * 4) {
        f16x4 t = ld_f16x4(te + i);
        float4 pv = *reinterpret_cast<const float4*>(pe + i);
        *reinterpret_cast<float4*>(x + (long)r * d + i) =
            make_float4((float)t[0] + pv.x, (float)t[1] + pv.y, (float)t[2] + pv.z, (float)t[3] + pv.w);
    }
    WLX_TR_END(trc);
}

void launch_dec_embed(const half_t* tok_emb, const float* pos_emb, int d, const RowTables& rt, int rows,
                      float* x, const int* done, hipStream_t s) {
    hipLaunchKernelGGL(dec_embed_kernel, dim3(rows), dim3(256), 0, s, tok_emb, pos_emb, d, rt.token, rt.pos,
                       rt.cache, rt.intok, x, done WLX_TR_ARG("embed"));
}

// ------------------------------------------------------------------ skinny GEMM ("GEMV") with fused prologue/epilogue
#define GV_CH 6   // k-tiles per register chunk

template <int MT, int NTB, int IN>
__global__ __launch_bounds__(512) void dec_gemv_kernel(GemvParams p) {
    if (p.done && *p.done) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int nw = blockDim.x >> 6;
    const int KT = p.KT;
    const int KTW = (KT + nw - 1) / nw;
    const int kt0 = wave * KTW;
    const int kt1 = (kt0 + KTW < KT) ? kt0 + KTW : KT;
    const int NT_total = (p.N + 15) >> 4;

    const half_t* wbase[NTB];
#pragma unroll
    for (int i = 0; i < NTB; ++i) {
        int nt = blockIdx.x * NTB + i;
        if (nt >= NT_total) nt = NT_total - 1;
        wbase[i] = p.Wp + ((long)nt * KT * 64 + lane) * 8;
    }

    f32x4 acc[NTB][MT];
#pragma unroll
    for (int i = 0; i < NTB; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[i][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- issue the first chunk of weight-fragment loads before anything else
    f16x8 wf[GV_CH][NTB];
#pragma unroll
    for (int j = 0; j < GV_CH; ++j) {
        int kt = kt0 + j;
        if (kt > KT - 1) kt = KT - 1;
#pragma unroll
        for (int i = 0; i < NTB; ++i) wf[j][i] = ld_f16x8(wbase[i] + (long)kt * 512);
    }

    f16x8 xf[GV_CH][MT];
    float* red = smem;                         // [2][nw][MT*16]
    float* accred = smem + 2 * nw * MT * 16;   // [nw][NTB*MT][64][4]

    if constexpr (IN == GEMV_IN_LN) {
        // LayerNorm over K = d_model of every live row, statistics shared through LDS.
        // (host guarantees KTW <= GV_CH in this mode: one chunk per wave)
        const float invK = 1.0f / (float)p.K;
        float mean[MT], rstd[MT];
        if constexpr (MT == 1) {
            float xr[GV_CH][8];
            const bool rowok = c < p.M;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < GV_CH; ++j) {
                const int kt = kt0 + j;
                if (kt < kt1 && rowok) {
                    const float4* xp = reinterpret_cast<const float4*>(p.X + (long)c * p.ldx + kt * 32 + g * 8);
                    float4 a = xp[0], b = xp[1];
                    xr[j][0] = a.x; xr[j][1] = a.y; xr[j][2] = a.z; xr[j][3] = a.w;
                    xr[j][4] = b.x; xr[j][5] = b.y; xr[j][6] = b.z; xr[j][7] = b.w;
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += xr[j][e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xr[j][e] = 0.f;
                }
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (g == 0) red[wave * 16 + c] = s;
            __syncthreads();
            float tot = 0.f;
            for (int w = 0; w < nw; ++w) tot += red[w * 16 + c];
            mean[0] = tot * invK;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < GV_CH; ++j) {
                if (kt0 + j < kt1 && rowok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { float dlt = xr[j][e] - mean[0]; q += dlt * dlt; }
                }
            }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (g == 0) red[nw * 16 + wave * 16 + c] = q;
            __syncthreads();
            float qt = 0.f;
            for (int w = 0; w < nw; ++w) qt += red[nw * 16 + w * 16 + c];
            rstd[0] = rsqrtf(qt * invK + 1e-5f);
#pragma unroll
            for (int j = 0; j < GV_CH; ++j) {
                const int kt = kt0 + j;
                f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                if (kt < kt1 && rowok) {
                    const float4* gp = reinterpret_cast<const float4*>(p.gamma + kt * 32 + g * 8);
                    const float4* bp = reinterpret_cast<const float4*>(p.beta + kt * 32 + g * 8);
                    float4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
                    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)((xr[j][e] - mean[0]) * rstd[0] * gg[e] + bb[e]);
                }
                xf[j][0] = o;
            }
        } else {
            // MT > 1 (prefill / batched rows): three passes over x (L1/L2 resident) instead of
            // holding MT*48 raw floats in registers.
            float s[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                s[mt] = 0.f;
                const int m = mt * 16 + c;
                if (m < p.M)
                    for (int kt = kt0; kt < kt1; ++kt) {
                        const float4* xp = reinterpret_cast<const float4*>(p.X + (long)m * p.ldx + kt * 32 + g * 8);
                        float4 a = xp[0], b = xp[1];
                        s[mt] += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
                    }
                s[mt] += __shfl_xor(s[mt], 16, 64);
                s[mt] += __shfl_xor(s[mt], 32, 64);
                if (g == 0) red[(wave * MT + mt) * 16 + c] = s[mt];
            }
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float tot = 0.f;
                for (int w = 0; w < nw; ++w) tot += red[(w * MT + mt) * 16 + c];
                mean[mt] = tot * invK;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float q = 0.f;
                const int m = mt * 16 + c;
                if (m < p.M)
                    for (int kt = kt0; kt < kt1; ++kt) {
                        const float4* xp = reinterpret_cast<const float4*>(p.X + (long)m * p.ldx + kt * 32 + g * 8);
                        float4 a = xp[0], b = xp[1];
                        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) { float dlt = v[e] - mean[mt]; q += dlt * dlt; }
                    }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                if (g == 0) red[nw * MT * 16 + (wave * MT + mt) * 16 + c] = q;
            }
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float qt = 0.f;
                for (int w = 0; w < nw; ++w) qt += red[nw * MT * 16 + (w * MT + mt) * 16 + c];
                rstd[mt] = rsqrtf(qt * invK + 1e-5f);
            }
#pragma unroll
            for (int j = 0; j < GV_CH; ++j) {
                const int kt = kt0 + j;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = mt * 16 + c;
                    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (kt < kt1 && m < p.M) {
                        const float4* xp = reinterpret_cast<const float4*>(p.X + (long)m * p.ldx + kt * 32 + g * 8);
                        const float4* gp = reinterpret_cast<const float4*>(p.gamma + kt * 32 + g * 8);
                        const float4* bp = reinterpret_cast<const float4*>(p.beta + kt * 32 + g * 8);
                        float4 a = xp[0], b = xp[1], g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
                        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (half_t)((v[e] - mean[mt]) * rstd[mt] * gg[e] + bb[e]);
                    }
                    xf[j][mt] = o;
                }
            }
        }
    }

    for (int base = kt0; base < kt1; base += GV_CH) {
        if constexpr (IN == GEMV_IN_F16) {
#pragma unroll
            for (int j = 0; j < GV_CH; ++j) {
                const int kt = base + j;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = mt * 16 + c;
                    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (kt < kt1 && m < p.M) o = ld_f16x8(p.Xh + (long)m * p.ldxh + kt * 32 + g * 8);
                    xf[j][mt] = o;
                }
            }
        } else if constexpr (IN == GEMV_IN_XATTN) {
            // combine the WLX_XSPLIT partials of the cross attention: per split a NORMALISED fp16 O row and fp32 (m, l)
#pragma unroll
            for (int j = 0; j < GV_CH; ++j) {
                const int kt = base + j;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = mt * 16 + c;
                    f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (kt < kt1 && m < p.M) {
                        const int k = kt * 32 + g * 8;
                        const int h = k >> 6, dd = k & 63;
                        const int item = m / p.R, qi = m - item * p.R;
                        const long ih = (long)item * p.H + h;
                        const float* mlp = p.part_ml + (ih * 16 + qi) * (WLX_XSPLIT * 2);
                        float wsp[WLX_XSPLIT];
                        float mmax = WLX_NEG_INF, den = 0.f;
#pragma unroll
                        for (int sp = 0; sp < WLX_XSPLIT; ++sp) mmax = fmaxf(mmax, mlp[sp * 2]);
#pragma unroll
                        for (int sp = 0; sp < WLX_XSPLIT; ++sp) { wsp[sp] = __expf(mlp[sp * 2] - mmax) * mlp[sp * 2 + 1]; den += wsp[sp]; }
                        float num[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int sp = 0; sp < WLX_XSPLIT; ++sp) {
                            const f16x8 ov = ld_f16x8(p.part_o + ((ih * WLX_XSPLIT + sp) * 16 + qi) * 64 + dd);
#pragma unroll
                            for (int e = 0; e < 8; ++e) num[e] += wsp[sp] * (float)ov[e];
                        }
                        const float inv = 1.0f / den;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (half_t)(num[e] * inv);
                    }
                    xf[j][mt] = o;
                }
            }
        }
        // prefetch the next chunk of weights (wave-uniform branch)
        f16x8 wn[GV_CH][NTB];
        const bool more = base + GV_CH < kt1;
        if (more) {
#pragma unroll
            for (int j = 0; j < GV_CH; ++j) {
                int kt = base + GV_CH + j;
                if (kt > KT - 1) kt = KT - 1;
#pragma unroll
                for (int i = 0; i < NTB; ++i) wn[j][i] = ld_f16x8(wbase[i] + (long)kt * 512);
            }
        }
#pragma unroll
        for (int j = 0; j < GV_CH; ++j) {
            if (base + j < kt1) {
#pragma unroll
                for (int i = 0; i < NTB; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma16(wf[j][i], xf[j][mt], acc[i][mt]);
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < GV_CH; ++j)
#pragma unroll
                for (int i = 0; i < NTB; ++i) wf[j][i] = wn[j][i];
        }
    }

    // ---- cross-wave K reduction through LDS
#pragma unroll
    for (int i = 0; i < NTB; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<f32x4*>(accred + (((long)wave * (NTB * MT) + (i * MT + mt)) * 64 + lane) * 4) = acc[i][mt];
    __syncthreads();

    for (int pair = wave; pair < NTB * MT; pair += nw) {
        const int i = pair / MT, mt = pair - i * MT;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < nw; ++w) {
            f32x4 t = *reinterpret_cast<const f32x4*>(accred + (((long)w * (NTB * MT) + pair) * 64 + lane) * 4);
            v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
        const int ntile = blockIdx.x * NTB + i;
        if (ntile >= NT_total) continue;
        const int n = ntile * 16 + g * 4;
        const int m = mt * 16 + c;
        if (m >= p.M) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = v[r] + ((p.bias && n + r < p.N) ? p.bias[n + r] : 0.f);
        switch (p.out_mode) {
            case GEMV_OUT_F16:
            case GEMV_OUT_GELU_F16: {
                if (p.out_mode == GEMV_OUT_GELU_F16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = gelu_erf(o[r]);
                }
                f16x4 h = {(half_t)(o[0] * p.qscale), (half_t)(o[1] * p.qscale),
                           (half_t)(o[2] * p.qscale), (half_t)(o[3] * p.qscale)};   // qscale = 1 unless a q projection
                *reinterpret_cast<f16x4*>(p.Yh + (long)m * p.ldyh + n) = h;
            } break;
            case GEMV_OUT_F32: {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) p.Y[(long)m * p.ldy + n + r] = o[r];
            } break;
            case GEMV_OUT_RESID: {
                float4* xp = reinterpret_cast<float4*>(p.Xres + (long)m * p.ldxres + n);
                float4 t = *xp;
                t.x += o[0]; t.y += o[1]; t.z += o[2]; t.w += o[3];
                *xp = t;
            } break;
            case GEMV_OUT_QKV: {
                if (n < p.d) {
                    f16x4 h = {(half_t)(o[0] * p.qscale), (half_t)(o[1] * p.qscale),
                               (half_t)(o[2] * p.qscale), (half_t)(o[3] * p.qscale)};
                    *reinterpret_cast<f16x4*>(p.Yh + (long)m * p.ldyh + n) = h;
                } else {
                    f16x4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                    const long off = (long)p.row_cache[m] * p.cache_row_stride + (long)p.row_pos[m] * p.d;
                    if (n < 2 * p.d) *reinterpret_cast<f16x4*>(p.Kc + off + (n - p.d)) = h;
                    else *reinterpret_cast<f16x4*>(p.Vc + off + (n - 2 * p.d)) = h;
                }
            } break;
            default: break;
        }
    }
}


bool g_decode_v1 = false;

#ifdef WLX_TRACE
unsigned long long* g_trace_buf = nullptr;
int g_trace_seq = 0;
const char* g_trace_names[512];
#endif

// ------------------------------------------------------------------ third generation: the LEAN skinny GEMM
// Measured on MI355X (scripts/ubench/chain3.hip): a dependent chain of 48-workgroup launches that stream
// 1.2 MB each costs 2.0 us per launch, and every KiB of straight-line code a launch executes adds ~0.4 us — the
// instruction cache is cold at every launch and cold code is fetched at ~3 GB/s, before or after the loads are issued.
// dec_gemv1_kernel is 5-7 KiB of fully unrolled, clamped, 64-bit-indexed code: 2-3 us of instruction fetch per launch,
// more than its HBM time. This kernel executes ~1 KiB:
//   * the wave's K slice is EXACT (host picks nw x CH x NCH == KT), so no clamps or predicates: loads are
//     base + immediate offset; rows >= M are never masked — MFMA output column j depends only on B column j, and
//     columns >= M are simply not stored;
//   * LayerNorm rows are reduced with DPP adds (12 VALU ops) instead of 12 dependent ds_bpermute round trips;
//   * everything rarely needed (ragged K, M > 16, d_model not a multiple of 256) stays in the older kernels.
__device__ __forceinline__ float wave_sum_dpp(float v) { return dpp_wave_sum(v); }   // common.h: 6 v_add_f32_dpp

// copy M rows x K fp16 (16-byte units) global -> LDS rows of stride ldxs, two units per thread in flight per trip (a rolled
// load->store loop pays one full L2 round trip per unit; M = 5 needs one trip for K = 768 / 256 threads and K = 3072 / 1024)
// `after_first_loads` runs once, between the first trip's global loads and its LDS stores (every thread runs it, also
// threads without a unit): the caller requests its weight stream there, behind the activation loads.
template <class F>
__device__ __forceinline__ void stage_rows_f16(const half_t* __restrict__ X, long ldx, int M, int K, half_t* xs, int ldxs,
                                               F&& after_first_loads) {
    const int kv8 = K >> 3, total = M * kv8, nthr = blockDim.x;
    {   // first trip, peeled: clamped units so every thread issues (and the hook sits at one program point)
        const int u0 = (threadIdx.x < total) ? (int)threadIdx.x : total - 1;
        const int u1 = (u0 + nthr < total) ? u0 + nthr : u0;
        const int m0 = u0 / kv8, k0 = u0 - m0 * kv8;
        const int m1 = u1 / kv8, k1 = u1 - m1 * kv8;
        const f16x8 v0 = ld_f16x8(X + (long)m0 * ldx + k0 * 8);
        const f16x8 v1 = ld_f16x8(X + (long)m1 * ldx + k1 * 8);
        after_first_loads();
        *reinterpret_cast<f16x8*>(xs + m0 * ldxs + k0 * 8) = v0;      // clamped duplicates rewrite a unit with its own value
        *reinterpret_cast<f16x8*>(xs + m1 * ldxs + k1 * 8) = v1;
    }
#pragma unroll 1
    for (int u0 = threadIdx.x + 2 * nthr; u0 < total; u0 += 2 * nthr) {
        const int u1 = u0 + nthr;
        const int m0 = u0 / kv8, k0 = u0 - m0 * kv8;
        const int uc = (u1 < total) ? u1 : u0;
        const int m1 = uc / kv8, k1 = uc - m1 * kv8;
        const f16x8 v0 = ld_f16x8(X + (long)m0 * ldx + k0 * 8);
        const f16x8 v1 = ld_f16x8(X + (long)m1 * ldx + k1 * 8);
        *reinterpret_cast<f16x8*>(xs + m0 * ldxs + k0 * 8) = v0;
        *reinterpret_cast<f16x8*>(xs + m1 * ldxs + k1 * 8) = v1;      // u1 out of range: rewrites unit u0 with its own value
    }
}

// The same for ONE WAVE's own K slice (round 6, WLX_STAGE_WAVE): wave w copies columns [0, K) of the M rows of ITS slice (X and xs already point at
// the slice) with its 64 lanes — the same number of load instructions per workgroup as the cooperative copy — and is the only reader of what it wrote,
// so the workgroup barrier between the staging and the MFMAs goes: a wave's LDS operations execute in order, and a wave starts its MFMAs when ITS
// loads have landed instead of when the slowest wave's have.
template <int P, class F>
__device__ __forceinline__ void stage_rows_f16_wave(const half_t* __restrict__ X, long ldx, int M, int K, half_t* xs, int ldxs, int lane,
                                                    F&& after_first_loads) {
    // P units per lane are requested together before anything is stored (P = what five rows of the slice need: one round trip); a unit index past
    // the end is clamped to the lane's previous unit, which is then stored twice with its own value
    const int kv8 = K >> 3, total = M * kv8;
    {
        int m_[P], k_[P];
        f16x8 v_[P];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            int u = lane + 64 * q;
            if (u >= total) u = (q == 0) ? total - 1 : (lane + 64 * (q - 1) < total ? lane + 64 * (q - 1) : total - 1);
            m_[q] = u / kv8; k_[q] = u - m_[q] * kv8;
            v_[q] = ld_f16x8(X + (long)m_[q] * ldx + k_[q] * 8);
        }
        after_first_loads();
#pragma unroll
        for (int q = 0; q < P; ++q) *reinterpret_cast<f16x8*>(xs + m_[q] * ldxs + k_[q] * 8) = v_[q];
    }
#pragma unroll 1
    for (int u0 = lane + 64 * P; u0 < total; u0 += 128) {
        const int u1 = u0 + 64;
        const int m0 = u0 / kv8, k0 = u0 - m0 * kv8;
        const int uc = (u1 < total) ? u1 : u0;
        const int m1 = uc / kv8, k1 = uc - m1 * kv8;
        const f16x8 v0 = ld_f16x8(X + (long)m0 * ldx + k0 * 8);
        const f16x8 v1 = ld_f16x8(X + (long)m1 * ldx + k1 * 8);
        *reinterpret_cast<f16x8*>(xs + m0 * ldxs + k0 * 8) = v0;
        *reinterpret_cast<f16x8*>(xs + m1 * ldxs + k1 * 8) = v1;
    }
}

// (Tried and dropped: a quarter-tile variant for fc2 — each 16-column tile shared by four workgroups, weights re-packed so
// a 1 KiB load holds four rows x four k-tiles, four MFMAs per load into per-lane-group accumulators, no cross-workgroup
// reduction. Numerically exact (all parity tests green) and it cuts the weight-load instructions per CU from 96 to 24, but
// every one of the 192 workgroups must stage the whole 5 x 3072 activation block: 5.5 us per launch vs 5.1.)

// Round 2, second half — three changes that each remove latency the decode-step trace showed (profiles/r2e_*):
//   * XS (GemvXsrc): the residual rows of the LayerNorm prologue / residual epilogue may be "rows + partial-sum slabs"
//     (GEMV_OUT_SLAB below) or, for layer 0, gathered from the embedding tables (the embedding launch is gone);
//   * GEMV_OUT_SLAB: the MLP output projection (K = 4 d_model: 96 KiB of weights and the whole 5 x 3072 activation block
//     per 16-column workgroup, on 48 CUs — the slowest launch of a layer, 4.8 us) is cut into WLX_FC2_KS K slices, grid
//     (tiles, slices); every slice workgroup writes its fp32 partial tile to its own slab and NOBODY reduces them in that
//     launch: the next layer's first projection sums rows + slabs in its LayerNorm prologue, and the next residual update
//     (the attention output projection) writes the sum back — a cross-workgroup reduction costs a launch boundary or a
//     grid barrier (>= 3 us either way), the deferred one costs WLX_FC2_KS more 1 KiB loads per row;
//   (Measured and dropped: LayerNorm helper waves beyond the nw MFMA waves, a wave per row — their dummy weight requests,
//   needed to keep hipcc's wait counting uniform, delayed the real weight stream by 0.6 us per launch, profiles/r2f_*.)
template <int CH, int LNV, int IN, int OUT, int NTB, int MT, int XS>
// (<= 8 waves wherever the kernel holds more than one row tile of fragments: 256 VGPRs per lane — at 16 waves the
// two- and three-tile residual projections spilled, 36-180 bytes of scratch per lane)
__global__ __launch_bounds__((IN == GEMV_IN_LN || OUT == GEMV_OUT_SLAB || MT > 1 || CH > 6 || (IN == GEMV_IN_XATTN && WLX_XCOMB_WAVE != 0)) ? 512 : 1024) void dec_gemv2_kernel(GemvParams p_in) {
    // Row chunks (prompt prefill, round 3): a pass over up to 448 rows runs every projection as ONE launch whose grid.z walks
    // chunks of 48 rows (three MFMA row tiles, the widest this kernel holds); a chunk is this kernel on rebased row pointers.
    // Decode steps launch with Mtot = 0 and skip the block (a scalar branch).
    // Row tiles (batched decode steps, round 4): 17..64 rows run as row chunks of ONE 16-row MFMA tile each (the MT = 1
    // instantiations — the ones tuned for a single stream), the chunk index folded into blockIdx.x so that the rt_nz
    // workgroups that stream the same weight tile are (a) on one XCD (ids 8 apart: one L2 fetches the tile from HBM once)
    // and (b) dispatched back to back: lin = ((tile / 8) * rt_nz + chunk) * 8 + tile % 8. The three-tile form (48 rows per
    // workgroup) made every one of the N / 16 workgroups normalise / stage ALL rows on N / 16 CUs (60 rows, d_model 768:
    // 7.1 us per residual projection at 0.04 of the HBM peak); row tiles spread the same work over 4x the CUs.
    GemvParams p = p_in;
    int tile = blockIdx.x;
    if (p_in.Mtot > 0) {
        int zc = (int)blockIdx.z;
        if (p_in.rt_nz > 0) {
            const int lin = (int)blockIdx.x, t = lin >> 3;
            const int tq = (int)(((unsigned)t * (unsigned)p_in.rt_magic) >> 16);      // t / rt_nz (host: magic = 65536 / nz + 1, exact for t < 32768)
            zc = t - tq * p_in.rt_nz;
            tile = tq * 8 + (lin & 7);
            if (tile >= p_in.rt_tiles) return;                                        // (the tile count is padded to a multiple of 8)
        }
        const int CHK = p_in.chunk;
        const int r0 = zc * CHK;
        p.M = (p_in.Mtot - r0 < CHK) ? p_in.Mtot - r0 : CHK;
        if (p.emb_token) p.emb_token += r0;
        if (p.X) p.X += (long)r0 * p.ldx;
        if (p.Xh) p.Xh += (long)r0 * p.ldxh;
        if (p.Yh) p.Yh += (long)r0 * p.ldyh;
        if (p.Y) p.Y += (long)r0 * p.ldy;
        if (p.Xres) p.Xres += (long)r0 * p.ldxres;
        if (p.slab) p.slab += (long)r0 * ((IN == GEMV_IN_LN) ? p.ldx : p.ldxres);
        if (p.row_cache) p.row_cache += r0;
        if (p.row_pos) p.row_pos += r0;
    }
    static_assert(XS == GEMV_X_PLAIN || IN == GEMV_IN_LN || (IN == GEMV_IN_F16 && OUT == GEMV_OUT_RESID && NTB == 1),
                  "slab / embedding sources: LayerNorm prologue or residual epilogue only");
    static_assert(OUT != GEMV_OUT_SLAB || (IN == GEMV_IN_F16 && NTB == 1), "K-split form: fp16 rows in");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // nw = waves that stream weights and run MFMAs; the LayerNorm prologue and the cross-attention combine may bring extra
    // waves that only help with the prologue (and load no weights: their wp is clamped to wave 0's slice, results unused)
    const int nw = p.nwm;
    WLX_TR_BEGIN();
    constexpr int NP = NTB * MT;                                           // (n-tile, 16-row tile) pairs of this workgroup
    // LayerNorm rows are held as NV float4 per lane, d_model = 256 NV. LNV == 15 stands for d_model 384 = 1.5 x 256 (tiny / tiny.en, round 5:
    // they ran on the first-generation kernel): two units, the second live on lanes 0..31 only — the other lanes load a clamped address,
    // hold zeros and store nothing. For every other LNV the masks below are compile-time constants and the code is what it was.
    constexpr bool LNT = (LNV == 15);
    constexpr int NV = LNT ? 2 : LNV;
    const bool tail_on = !LNT || lane < 32;
    const int tback = LNT ? (tail_on ? 0 : lane) : 0;                      // float4 units to step back in the last unit (inactive lanes read lane 0's)
    (void)tail_on; (void)tback;
    float* accred = smem;                                                  // [nw][NP][64][4]
    half_t* xs = reinterpret_cast<half_t*>(smem + nw * NP * 256);          // fp16 activation rows

    // (Tried and dropped: sub-tile workgroups — a 16-column tile shared by 2-4 workgroups, each streaming a quarter of
    // the weight rows with the other lanes masked. It spreads N = 768 layers over 192 CUs but does not reduce the number
    // of load INSTRUCTIONS a CU issues, which is what bounds these launches (~11 ns per wave-level load): no gain.)
    const bool streams = (IN != GEMV_IN_XATTN) || wave < nw;               // this wave streams weights and runs MFMAs (helper waves: XATTN only)
    const int kx0 = (streams ? wave : 0) * p.KTW;                          // first k-tile of this wave inside its K slice
    const int ks0 = (OUT == GEMV_OUT_SLAB) ? (int)blockIdx.y * p.KTS : 0;  // first k-tile of this workgroup's K slice
    const int kw0 = ks0 + kx0;                                             // ... of this wave, inside the weight matrix
    const half_t* wp = p.Wp + ((long)(tile * NTB) * p.KT + kw0) * 512 + lane * 8;
    const long wstep = (long)p.KT * 512;                                   // next n-tile
    f16x8 wf[CH][NTB];
    // The weight stream is requested AFTER the activation loads have been issued (round 2; -DWLX_X_FIRST=0 restores the
    // first order for A/B): vmcnt retires in order, so with the weights first the wave's wait for its few activation
    // loads (L2) was a wait for its whole weight slice (HBM) as well, and the LayerNorm / staging / combine that must
    // precede the MFMAs started only once the weights had landed (decode-step trace: "LN done" 1.4 us into a 2.2 us
    // launch). With the activations first their wait is vmcnt(#weight loads): the prologue runs under the weight stream.
    auto load_weights = [&]() {
        // compile-time fence: hipcc otherwise hoists these address-independent loads back above the activation loads
        asm volatile("" ::: "memory");
        // helper waves of the combine (wave >= nw, they leave before the MFMAs) issue the same NUMBER of loads, all of one
        // already-requested KiB: a branch around the loads would make hipcc count the waits that follow for the path
        // WITHOUT weights in flight, i.e. drain the weight stream inside the combine on the waves that do have it
        const half_t* wq = streams ? wp : p.Wp + ((long)(tile * NTB) * p.KT + ks0) * 512 + lane * 8;   // (helpers: this workgroup's own first KiB — one shared line for every workgroup's helpers was a hot spot in L2)
        const long js = streams ? 512 : 0, is = streams ? wstep : 0;
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int i = 0; i < NTB; ++i) wf[j][i] = ld_nt_f16x8(wq + i * is + j * js);
    };
#ifndef WLX_X_FIRST
#define WLX_X_FIRST 1
#endif
    if (!WLX_X_FIRST) load_weights();

    // epilogue operands of the FIRST pair this wave finishes (pair = wave: n-tile pair / MT, row tile pair % MT),
    // requested now (every lane, clamped row: no branch around a load); further pairs (batched rows) load theirs late
    int crow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) crow[mt] = (mt * 16 + c < p.M) ? mt * 16 + c : p.M - 1;
    const int pair0 = (wave < NP) ? wave : 0;
    const int nt_e = tile * NTB + pair0 / MT;
    const int n_e = nt_e * 16 + g * 4;
    int row_e = (pair0 % MT) * 16 + c;
    if (row_e >= p.M) row_e = p.M - 1;
    float4 bias_e = make_float4(0.f, 0.f, 0.f, 0.f), res_e = bias_e;
    int rc_e = 0, rp_e = 0;
    if constexpr (OUT != GEMV_OUT_F32) bias_e = *reinterpret_cast<const float4*>(p.bias + n_e);
    if constexpr (OUT == GEMV_OUT_RESID) res_e = *reinterpret_cast<const float4*>(p.Xres + (long)row_e * p.ldxres + n_e);
    if constexpr (OUT == GEMV_OUT_QKV) { rc_e = p.row_cache[row_e]; rp_e = p.row_pos[row_e]; }
    float4 slab_e[(OUT == GEMV_OUT_RESID && XS == GEMV_X_SLABS) ? WLX_FC2_KS : 1];
    if constexpr (OUT == GEMV_OUT_RESID && XS == GEMV_X_SLABS) {           // the residual is rows + slabs (summed at the store)
#pragma unroll
        for (int sl = 0; sl < WLX_FC2_KS; ++sl)
            slab_e[sl] = *reinterpret_cast<const float4*>(p.slab + sl * p.slab_stride + (long)row_e * p.ldxres + n_e);
    }
    if constexpr (OUT == GEMV_OUT_SLAB) { if (blockIdx.y != 0) bias_e = make_float4(0.f, 0.f, 0.f, 0.f); }   // the bias once: slice 0

    f32x4 acc[NTB][MT];
#pragma unroll
    for (int i = 0; i < NTB; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[i][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 xf[CH][MT];

    if constexpr (IN == GEMV_IN_F16) {
        const half_t* xr[MT];
        int xstep;                                                          // halfs between k-tiles of a row
        if (MT == 1 && p.xstage) {
            // One stream (M <= 16): the fp16 rows go through LDS — fetching B fragments straight from global costs CH
            // loads per wave of 64-byte pieces (16 waves x 6 = 96 load instructions per workgroup for K = 3072, as many
            // as the weights; a CU retires one per ~11 ns), the cooperative copy M * K / 8 / 64 = 30.
            const int Ks = (OUT == GEMV_OUT_SLAB) ? p.KTS * 32 : p.K;      // columns of the rows this workgroup multiplies
            const int ldxs = Ks + 8;
            if constexpr (WLX_STAGE_WAVE != 0) {
                // every wave stages and reads only its own K slice: no workgroup barrier (stage_rows_f16_wave)
                stage_rows_f16_wave<(CH * 4 * 5 + 63) / 64>(p.Xh + (ks0 + kx0) * 32, p.ldxh, p.M, p.KTW * 32, xs + kx0 * 32, ldxs, lane, [&]() { if (WLX_X_FIRST) load_weights(); });
                WLX_TR_MARK(1);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
            stage_rows_f16(p.Xh + ks0 * 32, p.ldxh, p.M, Ks, xs, ldxs, [&]() { if (WLX_X_FIRST) load_weights(); });
            WLX_TR_MARK(1);
            __syncthreads();
            }
            xr[0] = xs + crow[0] * ldxs + kx0 * 32 + g * 8;                 // lanes of rows >= M re-read a valid row (never stored)
            xstep = 32;
        } else {
            // batched rows, or K too large for the LDS budget (large-v3 fc2: 5 x 5120 fp16 + partials > 64 KiB):
            // fragments straight from global
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xr[mt] = p.Xh + (long)crow[mt] * p.ldxh + kw0 * 32 + g * 8;
            xstep = 32;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[j][mt] = *reinterpret_cast<const f16x8*>(xr[mt] + j * xstep);
        if (WLX_X_FIRST && !(MT == 1 && p.xstage)) load_weights();          // (staged form: requested inside stage_rows_f16)
#pragma unroll 1
        for (int ch = 1; ch < p.NCH; ++ch) {                               // big-K layers of the larger models only
            f16x8 wn[CH][NTB], xn[CH][MT];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
#pragma unroll
                for (int i = 0; i < NTB; ++i) wn[j][i] = ld_nt_f16x8(wp + i * wstep + (ch * CH + j) * 512);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xn[j][mt] = *reinterpret_cast<const f16x8*>(xr[mt] + (ch * CH + j) * xstep);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
#pragma unroll
                for (int i = 0; i < NTB; ++i) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma16(wf[j][i], xf[j][mt], acc[i][mt]);
                    wf[j][i] = wn[j][i];
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xf[j][mt] = xn[j][mt];
            }
        }
    } else if constexpr (IN == GEMV_IN_LN && XS != GEMV_X_PLAIN) {
        // rows from slabs or the embedding tables: ONE row per wave in the first trip (its pieces are 3-5x the registers of a
        // plain row; the host picks a K split with at least as many waves as rows, all of them streaming weights — helper waves
        // with clamped dummy loads were measured 0.6 us SLOWER per launch, profiles/r2f_*) — except the four-wave shapes of
        // round 6 (PF2 below: K = 1024 / 1280), which request the wave's second row together with its first.
        // wave w normalises rows w, w + nw, ...; a row = LNV float4 per lane (d_model = 256 LNV).
        const int nwl = nw;
        constexpr int NSL = (XS == GEMV_X_SLABS) ? WLX_FC2_KS : 1;
        float4 x[NV], sl[NSL][NV];
        f16x4 te[NV];
        int tok0 = 0, pos0 = 0;
        // request the pieces of row r (wave-uniform, clamped by the caller): rows, + slabs, or embedding + position
        auto request_row = [&](int r, float4 (&x)[NV], float4 (&sl)[NSL][NV], f16x4 (&te)[NV], int& tok, int& pos) {
            if constexpr (XS == GEMV_X_EMBED) {
                tok = p.emb_token[r];
                // (position and cache row in one word: both are scalar loads here — a vector load inside the lane-0 store
                // branch below would make hipcc drain the whole weight stream in front of it)
                pos = p.row_pos[r] | (p.row_cache[r] << 16);               // position < 448, cache row < 32768
                const half_t* tp = p.tok_emb + (long)tok * p.K + lane * 4;
                const float4* pp = reinterpret_cast<const float4*>(p.pos_emb + (long)(pos & 0xffff) * p.K) + lane;
#pragma unroll
                for (int j = 0; j < NV; ++j) { const int tb = (j == NV - 1) ? tback : 0; te[j] = ld_f16x4(tp + 256 * j - 4 * tb); x[j] = pp[64 * j - tb]; }
            } else {
                const float4* x4 = reinterpret_cast<const float4*>(p.X + (long)r * p.ldx) + lane;
#pragma unroll
                for (int j = 0; j < NV; ++j) x[j] = x4[64 * j - ((j == NV - 1) ? tback : 0)];
                if constexpr (XS == GEMV_X_SLABS) {
#pragma unroll
                    for (int q = 0; q < NSL; ++q) {
                        const float4* s4 = reinterpret_cast<const float4*>(p.slab + q * p.slab_stride + (long)r * p.ldx) + lane;
#pragma unroll
                        for (int j = 0; j < NV; ++j) sl[q][j] = s4[64 * j - ((j == NV - 1) ? tback : 0)];
                    }
                }
            }
        };
        // the row itself from its pieces (same association as the residual epilogue: ((x + s0) + s1) ...)
        auto combine_row = [&](int r, bool keep, float4 (&x)[NV], const float4 (&sl)[NSL][NV], const f16x4 (&te)[NV], int tok, int pos) {
            if constexpr (XS == GEMV_X_SLABS) {
#pragma unroll
                for (int q = 0; q < NSL; ++q)
#pragma unroll
                    for (int j = 0; j < NV; ++j) { x[j].x += sl[q][j].x; x[j].y += sl[q][j].y; x[j].z += sl[q][j].z; x[j].w += sl[q][j].w; }
            }
            if constexpr (XS == GEMV_X_EMBED) {
#pragma unroll
                for (int j = 0; j < NV; ++j) { x[j].x += (float)te[j][0]; x[j].y += (float)te[j][1]; x[j].z += (float)te[j][2]; x[j].w += (float)te[j][3]; }
                if (tile == 0 && keep) {            // workgroup 0 (of its row chunk) leaves the rows where the residual updates expect them
                    float4* o4 = reinterpret_cast<float4*>(p.Xres + (long)r * p.ldxres) + lane;
#pragma unroll
                    for (int j = 0; j < NV; ++j) if (j < NV - 1 || tail_on) o4[64 * j] = x[j];
                    if (lane == 0) p.intok[(long)(pos >> 16) * WLX_T_TEXT + (pos & 0xffff)] = tok;
                }
            }
        };
        const int ra = (wave < p.M) ? wave : p.M - 1;
        request_row(ra, x, sl, te, tok0, pos0);
        // The four-wave shapes of one stream's step (log G9: (CH, LNV) = (6, 3), (8, 4), (10, 5) — instantiated for nothing else) have fewer waves than
        // rows: the wave's SECOND row is requested together with its first, as the PLAIN-rows prologue below does — a clamped row for the waves that
        // have none (every wave issues the same loads: no branch around a request)
        constexpr bool PF2 = MT == 1 && ((CH == 6 && LNV == 3) || (CH == 8 && LNV == 4) || (CH == 10 && LNV == 5));
        float4 xb[NV], slb[NSL][NV];
        f16x4 teb[NV];
        int tokb = 0, posb = 0;
        const int rb = (wave + nwl < p.M) ? wave + nwl : p.M - 1;
        if constexpr (PF2) request_row(rb, xb, slb, teb, tokb, posb);
        const float4* g4 = reinterpret_cast<const float4*>(p.gamma) + lane;
        const float4* b4 = reinterpret_cast<const float4*>(p.beta) + lane;
        float4 gq[NV], bq[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { const int tb = (j == NV - 1) ? tback : 0; gq[j] = g4[64 * j - tb]; bq[j] = b4[64 * j - tb]; }
        if (WLX_X_FIRST) load_weights();
        const int ldxs = p.K + 8;
        constexpr float invK = LNT ? (1.0f / 384.0f) : 1.0f / (256.0f * NV);
        auto ln_row = [&](float4 (&x)[NV], int r, bool keep) {
            if constexpr (LNT) { if (!tail_on) x[NV - 1] = make_float4(0.f, 0.f, 0.f, 0.f); }
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) sm += (x[j].x + x[j].y) + (x[j].z + x[j].w);
            const float mean = wave_sum_dpp(sm) * invK;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                x[j].x -= mean; x[j].y -= mean; x[j].z -= mean; x[j].w -= mean;
                if constexpr (LNT) { if (j == NV - 1 && !tail_on) x[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
                q += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
            }
            const float rstd = rsqrtf(wave_sum_dpp(q) * invK + 1e-5f);
            half_t* dst = xs + (long)r * ldxs + lane * 4;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const f16x4 hv = {(half_t)(x[j].x * rstd * gq[j].x + bq[j].x), (half_t)(x[j].y * rstd * gq[j].y + bq[j].y),
                                  (half_t)(x[j].z * rstd * gq[j].z + bq[j].z), (half_t)(x[j].w * rstd * gq[j].w + bq[j].w)};
                if (keep && (j < NV - 1 || tail_on)) *reinterpret_cast<f16x4*>(dst + 256 * j) = hv;
            }
        };
        // first trip: straight-line and UNCONDITIONAL (a wave without a row normalises the clamped row it loaded and keeps
        // nothing): inside an `if (wave < M)` hipcc sinks the row loads into the branch, behind the weights. It must not
        // share a loop with the later trips either: hipcc's wait insertion merges the two ways into a loop body by the
        // NEWEST request of either, so the first trip would wait for the weight stream it is meant to overlap.
        combine_row(ra, wave < p.M, x, sl, te, tok0, pos0);
        ln_row(x, ra, wave < p.M);
        if constexpr (PF2) {
            combine_row(rb, wave + nwl < p.M, xb, slb, teb, tokb, posb);
            ln_row(xb, rb, wave + nwl < p.M);
        }
        if constexpr (MT == 1) {
#pragma unroll 1
            for (int r = wave + (PF2 ? 2 : 1) * nwl; r < p.M; r += nwl) {   // more rows than waves (9..16 rows)
                float4 x2[NV], sl2[NSL][NV];
                f16x4 te2[NV];
                int tok2 = 0, pos2 = 0;
                request_row(r, x2, sl2, te2, tok2, pos2);
                combine_row(r, true, x2, sl2, te2, tok2, pos2);
                ln_row(x2, r, true);
            }
        } else {
            // batched streams (17..48 rows: 3..6 rows per wave): two rows per trip, both requested before either is
            // normalised, so a trip pays ONE round trip to L2 instead of one per row
#pragma unroll 1
            for (int r = wave + nwl; r < p.M; r += 2 * nwl) {
                const int r1 = r + nwl;
                const bool has1 = r1 < p.M;
                float4 x2[NV], sl2[NSL][NV], x3[NV], sl3[NSL][NV];
                f16x4 te2[NV], te3[NV];
                int tok2 = 0, pos2 = 0, tok3 = 0, pos3 = 0;
                request_row(r, x2, sl2, te2, tok2, pos2);
                request_row(has1 ? r1 : r, x3, sl3, te3, tok3, pos3);
                combine_row(r, true, x2, sl2, te2, tok2, pos2);
                ln_row(x2, r, true);
                combine_row(has1 ? r1 : r, has1, x3, sl3, te3, tok3, pos3);
                ln_row(x3, has1 ? r1 : r, has1);
            }
        }
        WLX_TR_MARK(1);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const half_t* xr = xs + crow[mt] * ldxs + kx0 * 32 + g * 8;     // lanes of rows >= M re-read a valid row (never stored)
#pragma unroll
            for (int j = 0; j < CH; ++j) xf[j][mt] = *reinterpret_cast<const f16x8*>(xr + j * 32);
        }
    } else if constexpr (IN == GEMV_IN_LN) {
        // wave w normalises rows w, w + nw, ...; a row = LNV float4 per lane (d_model = 256 LNV)
        // first trip's rows (this wave's row and the one nw below it) are requested before anything else
        float4 x[NV], y[NV];
        {
            const int ra = (wave < p.M) ? wave : p.M - 1, rb = (wave + nw < p.M) ? wave + nw : ra;
            const float4* x4 = reinterpret_cast<const float4*>(p.X + (long)ra * p.ldx) + lane;
            const float4* y4 = reinterpret_cast<const float4*>(p.X + (long)rb * p.ldx) + lane;
#pragma unroll
            for (int j = 0; j < NV; ++j) { const int tb = (j == NV - 1) ? tback : 0; x[j] = x4[64 * j - tb]; y[j] = y4[64 * j - tb]; }
        }
        const float4* g4 = reinterpret_cast<const float4*>(p.gamma) + lane;
        const float4* b4 = reinterpret_cast<const float4*>(p.beta) + lane;
        float4 gq[NV], bq[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { const int tb = (j == NV - 1) ? tback : 0; gq[j] = g4[64 * j - tb]; bq[j] = b4[64 * j - tb]; }
        if (WLX_X_FIRST) load_weights();
        const int ldxs = p.K + 8;
        constexpr float invK = LNT ? (1.0f / 384.0f) : 1.0f / (256.0f * NV);
        auto ln_row = [&](float4 (&x)[NV], int r) {
            if constexpr (LNT) { if (!tail_on) x[NV - 1] = make_float4(0.f, 0.f, 0.f, 0.f); }
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) sm += (x[j].x + x[j].y) + (x[j].z + x[j].w);
            const float mean = wave_sum_dpp(sm) * invK;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                x[j].x -= mean; x[j].y -= mean; x[j].z -= mean; x[j].w -= mean;
                if constexpr (LNT) { if (j == NV - 1 && !tail_on) x[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
                q += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
            }
            const float rstd = rsqrtf(wave_sum_dpp(q) * invK + 1e-5f);
            half_t* dst = xs + (long)r * ldxs + lane * 4;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const f16x4 hv = {(half_t)(x[j].x * rstd * gq[j].x + bq[j].x), (half_t)(x[j].y * rstd * gq[j].y + bq[j].y),
                                  (half_t)(x[j].z * rstd * gq[j].z + bq[j].z), (half_t)(x[j].w * rstd * gq[j].w + bq[j].w)};
                if (j < NV - 1 || tail_on) *reinterpret_cast<f16x4*>(dst + 256 * j) = hv;
            }
        };
        // first trip, straight-line: its rows were requested at the top. (It must not share a loop with the later trips:
        // hipcc's wait insertion merges the two ways into a loop body by the NEWEST request of either, so a body that
        // also re-loads x / y for a later trip makes the first trip wait for all but five of everything outstanding —
        // i.e. for the weight stream the reordering above is meant to overlap.)
        if (wave < p.M) {
            const bool has1 = wave + nw < p.M;
#pragma unroll 1
            for (int u = 0; u < (has1 ? 2 : 1); ++u) {                      // rolled: one copy of the row code (code size is latency here)
                if (u) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) x[j] = y[j];
                }
                ln_row(x, u ? wave + nw : wave);
            }
        }
#pragma unroll 1
        for (int r = wave + 2 * nw; r < p.M; r += 2 * nw) {                 // batched rows (M > 2 nw): rows r and r + nw per trip
            const int r1 = r + nw;
            const bool has1 = r1 < p.M;
            const float4* x4 = reinterpret_cast<const float4*>(p.X + (long)r * p.ldx) + lane;
            const float4* y4 = reinterpret_cast<const float4*>(p.X + (long)(has1 ? r1 : r) * p.ldx) + lane;
            float4 x2[NV], y2[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) { const int tb = (j == NV - 1) ? tback : 0; x2[j] = x4[64 * j - tb]; y2[j] = y4[64 * j - tb]; }
#pragma unroll 1
            for (int u = 0; u < (has1 ? 2 : 1); ++u) {
                if (u) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) x2[j] = y2[j];
                }
                ln_row(x2, u ? r1 : r);
            }
        }
        WLX_TR_MARK(1);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const half_t* xr = xs + crow[mt] * ldxs + kx0 * 32 + g * 8;     // lanes of rows >= M re-read a valid row (never stored)
#pragma unroll
            for (int j = 0; j < CH; ++j) xf[j][mt] = *reinterpret_cast<const f16x8*>(xr + j * 32);
        }
    } else {   // GEMV_IN_XATTN: every wave of the workgroup (more than the nw MFMA waves) combines the WLX_XSPLIT
        // partials (normalised fp16 O, fp32 (m, l) contiguous per row): one (row, head, 8-dim group) per thread, 12 16-byte
        // loads in flight, ONE L2 round trip when M * H * 8 <= blockDim; the fp16 result rows go to LDS like the LayerNorm
        // rows. (A CU retires one wave-level load instruction per ~11 ns: the fp32 / 4-dim / separate-(m,l) form needed
        // 240 of them per workgroup, this one 90.)
        const int ldxs = p.K + 8;
        const int n_it = p.M * p.H * 8;
        const float rH = 1.0f / (float)p.H, rR = 1.0f / (float)p.R;
        // One (row, head, 8-dim group) per call. The FIRST trip is straight-line code run by every thread (clamped item,
        // result not stored when out of range) with the weight request right behind its loads; later trips (M * H * 8 >
        // blockDim: batched rows) run in a separate loop — sharing one loop would make hipcc wait for the weights inside
        // the first trip (its wait insertion merges loop paths by the newest request of either).
        auto combine = [&](int it0, bool first) {
            const int it = (it0 < n_it) ? it0 : n_it - 1;
            const int q8 = it & 7, hm = it >> 3;
            const int m = (int)(((float)hm + 0.5f) * rH), hh = hm - m * p.H;
            const int item = (int)(((float)m + 0.5f) * rR), qi = m - item * p.R;
            const long ih = (long)item * p.H + hh;
            const float4* mlp = reinterpret_cast<const float4*>(p.part_ml + (ih * 16 + qi) * (WLX_XSPLIT * 2));
            const half_t* op = p.part_o + (ih * WLX_XSPLIT * 16 + qi) * 64 + q8 * 8;
            float4 ml[WLX_XSPLIT / 2];
            f16x8 ov[WLX_XSPLIT];
#pragma unroll
            for (int sp = 0; sp < WLX_XSPLIT / 2; ++sp) ml[sp] = mlp[sp];               // (m, l) of splits 2 sp, 2 sp + 1
#pragma unroll
            for (int sp = 0; sp < WLX_XSPLIT; ++sp) ov[sp] = ld_f16x8(op + sp * 1024);
            if (first && WLX_X_FIRST) load_weights();                       // behind the first trip's partial loads
            float mmax = fmaxf(ml[0].x, ml[0].z);
#pragma unroll
            for (int sp = 1; sp < WLX_XSPLIT / 2; ++sp) mmax = fmaxf(mmax, fmaxf(ml[sp].x, ml[sp].z));
            float den = 0.f;
            float num[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sp = 0; sp < WLX_XSPLIT; ++sp) {
                const float mm = (sp & 1) ? ml[sp >> 1].z : ml[sp >> 1].x, ll = (sp & 1) ? ml[sp >> 1].w : ml[sp >> 1].y;
                const float w = __expf(mm - mmax) * ll;
                den += w;
#pragma unroll
                for (int e = 0; e < 8; ++e) num[e] += w * (float)ov[sp][e];
            }
            const float inv = 1.0f / den;
            const f16x8 hv = {(half_t)(num[0] * inv), (half_t)(num[1] * inv), (half_t)(num[2] * inv), (half_t)(num[3] * inv),
                              (half_t)(num[4] * inv), (half_t)(num[5] * inv), (half_t)(num[6] * inv), (half_t)(num[7] * inv)};
            if (it0 < n_it) *reinterpret_cast<f16x8*>(xs + m * ldxs + hh * 64 + q8 * 8) = hv;
        };
        if constexpr (WLX_XCOMB_WAVE != 0 && MT == 1) {
            // Round 6 (WLX_XCOMB_WAVE; 0 = the cooperative form below, for A/B): every wave combines exactly the columns of ITS K slice — items
            // (row, 8-dim group) of columns [c0, c0 + W) — and is their only reader, so neither the helper waves nor the workgroup barrier between
            // the combine and the MFMAs are needed (the launcher starts nw waves): a wave goes on when ITS partials have landed. The first two
            // items of a lane are requested together, the weights right behind them (5 rows x 192 columns = 120 items: one round trip).
            const int W = p.KTW * 32, c0 = kx0 * 32, wg8 = W >> 3;
            const int n_w = p.M * wg8;
            const float rG = 1.0f / (float)wg8;
            auto item_ptrs = [&](int u, const float4*& mlp, const half_t*& op, int& m, int& col) {
                const int uu = (u < n_w) ? u : n_w - 1;
                m = (int)(((float)uu + 0.5f) * rG);                         // uu / wg8 (exact: small integers)
                col = c0 + (uu - m * wg8) * 8;
                const int hh = col >> 6, q8 = (col >> 3) & 7;
                const int item = (int)(((float)m + 0.5f) * rR), qi = m - item * p.R;
                const long ih = (long)item * p.H + hh;
                mlp = reinterpret_cast<const float4*>(p.part_ml + (ih * 16 + qi) * (WLX_XSPLIT * 2));
                op = p.part_o + (ih * WLX_XSPLIT * 16 + qi) * 64 + q8 * 8;
            };
            auto finish = [&](const float4 (&ml)[WLX_XSPLIT / 2], const f16x8 (&ov)[WLX_XSPLIT], int m, int col, bool keep) {
                float mmax = fmaxf(ml[0].x, ml[0].z);
#pragma unroll
                for (int sp = 1; sp < WLX_XSPLIT / 2; ++sp) mmax = fmaxf(mmax, fmaxf(ml[sp].x, ml[sp].z));
                float den = 0.f;
                float num[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int sp = 0; sp < WLX_XSPLIT; ++sp) {                   // (the arithmetic and its order are the cooperative form's: identical rows)
                    const float mm = (sp & 1) ? ml[sp >> 1].z : ml[sp >> 1].x, ll = (sp & 1) ? ml[sp >> 1].w : ml[sp >> 1].y;
                    const float w = __expf(mm - mmax) * ll;
                    den += w;
#pragma unroll
                    for (int e = 0; e < 8; ++e) num[e] += w * (float)ov[sp][e];
                }
                const float inv = 1.0f / den;
                const f16x8 hv = {(half_t)(num[0] * inv), (half_t)(num[1] * inv), (half_t)(num[2] * inv), (half_t)(num[3] * inv),
                                  (half_t)(num[4] * inv), (half_t)(num[5] * inv), (half_t)(num[6] * inv), (half_t)(num[7] * inv)};
                if (keep) *reinterpret_cast<f16x8*>(xs + m * ldxs + col) = hv;
            };
            {
                // NPL items per lane requested together (what five rows of the slice need: 2 for six k-tiles per wave, 3 for eight — one round trip)
                constexpr int NPL = (CH * 4 * 5 + 63) / 64 < 2 ? 2 : (CH * 4 * 5 + 63) / 64;
                const float4* mlpq[NPL]; const half_t* opq[NPL]; int mq[NPL], colq[NPL];
                float4 mlq[NPL][WLX_XSPLIT / 2];
                f16x8 ovq[NPL][WLX_XSPLIT];
#pragma unroll
                for (int q = 0; q < NPL; ++q) item_ptrs(lane + 64 * q, mlpq[q], opq[q], mq[q], colq[q]);
#pragma unroll
                for (int sp = 0; sp < WLX_XSPLIT / 2; ++sp)
#pragma unroll
                    for (int q = 0; q < NPL; ++q) mlq[q][sp] = mlpq[q][sp];
#pragma unroll
                for (int sp = 0; sp < WLX_XSPLIT; ++sp)
#pragma unroll
                    for (int q = 0; q < NPL; ++q) ovq[q][sp] = ld_f16x8(opq[q] + sp * 1024);
                if (WLX_X_FIRST) load_weights();
#pragma unroll
                for (int q = 0; q < NPL; ++q) finish(mlq[q], ovq[q], mq[q], colq[q], lane + 64 * q < n_w);
            }
#pragma unroll 1
            for (int u = lane + 64 * ((CH * 4 * 5 + 63) / 64 < 2 ? 2 : (CH * 4 * 5 + 63) / 64); u < n_w; u += 64) {   // more items per wave than the peeled ones (batched rows up to 16)
                const float4* mlp; const half_t* op; int m, col;
                item_ptrs(u, mlp, op, m, col);
                float4 ml[WLX_XSPLIT / 2];
                f16x8 ov[WLX_XSPLIT];
#pragma unroll
                for (int sp = 0; sp < WLX_XSPLIT / 2; ++sp) ml[sp] = mlp[sp];
#pragma unroll
                for (int sp = 0; sp < WLX_XSPLIT; ++sp) ov[sp] = ld_f16x8(op + sp * 1024);
                finish(ml, ov, m, col, true);
            }
            WLX_TR_MARK(1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
        combine(tid, true);
#pragma unroll 1
        for (int it0 = tid + blockDim.x; it0 < n_it; it0 += blockDim.x) combine(it0, false);
        WLX_TR_MARK(1);
        __syncthreads();
        if (!streams) return;                                               // helper waves are done (no later barrier needs them: ended waves leave the barrier count)
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const half_t* xr = xs + crow[mt] * ldxs + kx0 * 32 + g * 8;
#pragma unroll
            for (int j = 0; j < CH; ++j) xf[j][mt] = *reinterpret_cast<const f16x8*>(xr + j * 32);
        }
    }
    WLX_TR_MARK(2);
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int i = 0; i < NTB; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma16(wf[j][i], xf[j][mt], acc[i][mt]);

    // ---- cross-wave K reduction through LDS in a fixed order; wave w finishes pairs w, w + nw, ...
#pragma unroll
    for (int i = 0; i < NTB; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<f32x4*>(accred + ((wave * NP + i * MT + mt) * 64 + lane) * 4) = acc[i][mt];
    WLX_TR_MARK(3);
    __syncthreads();
#ifdef WLX_TRACE
    if (wave >= NP) { WLX_TR_END_WAVES(p.trc); return; }
#else
    if (wave >= NP) return;
#endif
#pragma unroll 1
    for (int pair = wave; pair < NP; pair += nw) {
    const int nt_p = tile * NTB + pair / MT;
    const int n_p = nt_p * 16 + g * 4;
    const int row_p = (pair % MT) * 16 + c;
    if (pair != wave) {                        // not the pair whose operands were requested up front
        const int rr = (row_p < p.M) ? row_p : p.M - 1;
        if constexpr (OUT != GEMV_OUT_F32) bias_e = *reinterpret_cast<const float4*>(p.bias + n_p);
        if constexpr (OUT == GEMV_OUT_RESID) res_e = *reinterpret_cast<const float4*>(p.Xres + (long)rr * p.ldxres + n_p);
        if constexpr (OUT == GEMV_OUT_RESID && XS == GEMV_X_SLABS) {
#pragma unroll
            for (int sl = 0; sl < WLX_FC2_KS; ++sl)
                slab_e[sl] = *reinterpret_cast<const float4*>(p.slab + sl * p.slab_stride + (long)rr * p.ldxres + n_p);
        }
        if constexpr (OUT == GEMV_OUT_SLAB) { if (blockIdx.y != 0) bias_e = make_float4(0.f, 0.f, 0.f, 0.f); }
        if constexpr (OUT == GEMV_OUT_QKV) { rc_e = p.row_cache[rr]; rp_e = p.row_pos[rr]; }
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* ar = accred + (pair * 64 + lane) * 4;
#pragma unroll 2
    for (int w = 0; w < nw; ++w) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(ar + w * (NP * 256));
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    WLX_TR_MARK(4);
    if (row_p < p.M && (NTB == 1 || nt_p * 16 < p.N)) {
        const int c = row_p;                   // (shadows the lane's column index: below, c is the activation row)
        const int n_e = n_p;
        float o0 = v[0] + bias_e.x, o1 = v[1] + bias_e.y, o2 = v[2] + bias_e.z, o3 = v[3] + bias_e.w;
        if constexpr (OUT == GEMV_OUT_F16 || OUT == GEMV_OUT_GELU_F16) {
            if constexpr (OUT == GEMV_OUT_GELU_F16) { o0 = gelu_erf(o0); o1 = gelu_erf(o1); o2 = gelu_erf(o2); o3 = gelu_erf(o3); }
            const f16x4 h = {(half_t)(o0 * p.qscale), (half_t)(o1 * p.qscale), (half_t)(o2 * p.qscale), (half_t)(o3 * p.qscale)};
            *reinterpret_cast<f16x4*>(p.Yh + (long)c * p.ldyh + n_e) = h;
        } else if constexpr (OUT == GEMV_OUT_F32) {
            float* yp = p.Y + (long)c * p.ldy + n_e;
            if (n_e + 3 < p.N) *reinterpret_cast<float4*>(yp) = make_float4(o0, o1, o2, o3);
            else { if (n_e < p.N) yp[0] = o0; if (n_e + 1 < p.N) yp[1] = o1; if (n_e + 2 < p.N) yp[2] = o2; }
        } else if constexpr (OUT == GEMV_OUT_RESID) {
            if constexpr (XS == GEMV_X_SLABS) {                             // rows + slabs (the LayerNorm prologue's association)
#pragma unroll
                for (int sl = 0; sl < WLX_FC2_KS; ++sl) { res_e.x += slab_e[sl].x; res_e.y += slab_e[sl].y; res_e.z += slab_e[sl].z; res_e.w += slab_e[sl].w; }
            }
            *reinterpret_cast<float4*>(p.Xres + (long)c * p.ldxres + n_e) =
                make_float4(res_e.x + o0, res_e.y + o1, res_e.z + o2, res_e.w + o3);
        } else if constexpr (OUT == GEMV_OUT_SLAB) {                        // this K slice's partial tile; summed by the consumers
            *reinterpret_cast<float4*>(p.slab + blockIdx.y * p.slab_stride + (long)c * p.ldxres + n_e) = make_float4(o0, o1, o2, o3);
        } else {   // GEMV_OUT_QKV: the 16-column tile lies entirely in q, k or v (d % 16 == 0)
            if (n_e < p.d) {
                const f16x4 h = {(half_t)(o0 * p.qscale), (half_t)(o1 * p.qscale), (half_t)(o2 * p.qscale), (half_t)(o3 * p.qscale)};
                *reinterpret_cast<f16x4*>(p.Yh + (long)c * p.ldyh + n_e) = h;
            } else {
                const f16x4 h = {(half_t)o0, (half_t)o1, (half_t)o2, (half_t)o3};
                const bool isk = n_e < 2 * p.d;
                half_t* dst = (isk ? p.Kc : p.Vc) + (long)rc_e * p.cache_row_stride + (long)rp_e * p.d + (n_e - (isk ? p.d : 2 * p.d));
                *reinterpret_cast<f16x4*>(dst) = h;
            }
        }
    }
    }
    WLX_TR_MARK(5);
    WLX_TR_END_WAVES(p.trc);
}

// one launch of an instantiation; workgroups that need more than the default 64 KiB of dynamic LDS (batched rows of the
// larger models: 30 rows x 1280 fp16 = 77 KiB of staged activations) raise the kernel's limit first, once. The first
// launch of every shape happens OUTSIDE stream capture (engine.hip runs a decoder pass eagerly before it captures one).
#define WLX_G2_LDS_MAX (152 * 1024)
// Set when a device refused the raised limit (another GPU generation, a lower per-block LDS limit): gemv2_cfg then keeps
// every shape that needs more than 64 KiB on the general kernel instead of launching something that cannot run.
static std::atomic<bool> g_lds_optin_refused{false};
template <int CH, int LNV, int IN, int OUT, int NTB, int MT, int XS>
static void g2_launch(dim3 grid, dim3 block, size_t shm, hipStream_t s, const GemvParams& p) {
    if (shm > 64 * 1024) {
        static std::atomic<signed char> granted[64] = {};   // per device: the opt-in is a property of the function ON a device
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && granted[dev].load(std::memory_order_acquire) == 0) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_gemv2_kernel<CH, LNV, IN, OUT, NTB, MT, XS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, WLX_G2_LDS_MAX);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                g_lds_optin_refused.store(true);
                fprintf(stderr, "[wlx] device %d refused %d KiB of dynamic LDS (%s): batched decode projections fall back to the general kernel\n",
                        dev, WLX_G2_LDS_MAX / 1024, hipGetErrorString(e));
                // the launch below then fails with the runtime's own error, which the engine's hipGetLastError check reports for
                // THIS call; later calls take the general kernel (gemv2_cfg)
            } else {
                granted[dev].store(1, std::memory_order_release);
            }
        }
    }
    hipLaunchKernelGGL((dec_gemv2_kernel<CH, LNV, IN, OUT, NTB, MT, XS>), grid, block, shm, s, p);
}

struct Gemv2Cfg { bool ok, xstage; int nw, CH, NCH, LNV, NTB, MT; size_t shm; };
static Gemv2Cfg gemv2_cfg(const GemvParams& p) {
    Gemv2Cfg c{};
    c.ok = false;
    if (g_decode_v1 || p.M > 48 || p.M < 1) return c;
    if (p.bias ? (p.N & 15) != 0 : p.out_mode != GEMV_OUT_F32) return c;      // bias <=> not the vocabulary projection
    const bool combo = (p.in_mode == GEMV_IN_LN && (p.out_mode == GEMV_OUT_QKV || p.out_mode == GEMV_OUT_F16 ||
                                                    p.out_mode == GEMV_OUT_GELU_F16 || p.out_mode == GEMV_OUT_F32)) ||
                       (p.in_mode != GEMV_IN_LN && p.out_mode == GEMV_OUT_RESID) ||
                       (p.in_mode == GEMV_IN_F16 && p.out_mode == GEMV_OUT_SLAB);
    if (!combo || p.K != p.KT * 32) return c;
    // sources other than the plain rows: one row tile, and only where the kernel is instantiated for them
    if (p.xsrc != GEMV_X_PLAIN) {
        const bool ln_ok = p.in_mode == GEMV_IN_LN && p.out_mode == GEMV_OUT_QKV;
        const bool res_ok = p.in_mode == GEMV_IN_F16 && p.out_mode == GEMV_OUT_RESID && p.xsrc == GEMV_X_SLABS;
        if (!(ln_ok || res_ok)) return c;
    }
    int KTf = p.KT;                                                           // k-tiles one workgroup multiplies
    if (p.out_mode == GEMV_OUT_SLAB) {
        if (p.KTS < 1 || p.KT % p.KTS || p.KT / p.KTS != WLX_FC2_KS) return c;
        KTf = p.KTS;
    }
    const int cap = (p.in_mode == GEMV_IN_F16 && p.out_mode != GEMV_OUT_SLAB && p.M <= 16) ? 16 : 8;
    // exact factorisation KTf = nw * CH * NCH, CH in {6, 5, 4}: fewest chunks first, then the widest chunk
    int best_nch = 1 << 30;
    // Log G9 (round 6): a layer's FIRST projection (rows + slabs / embedding rows) of one stream's step as FOUR waves for K = 1024 / 1280 (8 / 10 k-tiles
    // per wave instead of eight waves of 4 / 5), the wave's second row requested together with its first (PF2 in the kernel: without that prefetch
    // the four-wave shape LOST 0.6-1 %): large-v3 +1.6 %, medium.en +0.9 %, profiles/r6az_*. K = 768 keeps one row per wave (six waves of four: the
    // four-wave shape measured +0.1 %, inside the spread). WLX_G2_XS_FEW (A/B builds): 0 = one row per wave everywhere, 2 = K = 768 as four waves too.
    static const int xs_few = [] { const char* e = wlx_ab("WLX_G2_XS_FEW"); const int v = e ? atoi(e) : 1; return (v >= 0 && v <= 2) ? v : 1; }();
    const bool xs_few_here = xs_few && p.in_mode == GEMV_IN_LN && p.xsrc != GEMV_X_PLAIN && p.out_mode == GEMV_OUT_QKV && p.M <= 8 && p.Mtot == 0 &&
                             (p.K == 1024 || p.K == 1280 || (xs_few == 2 && p.K == 768));
    if (p.in_mode == GEMV_IN_LN && p.xsrc != GEMV_X_PLAIN && !xs_few_here) {
        // slab / embedding rows: one row per wave, so at least min(M, 8) waves, each streaming CH >= 2 k-tiles
        const int want = std::min(p.M, 8);
        for (int CH = 6; CH >= 2; --CH) {
            if (KTf % CH || KTf / CH > 8 || KTf / CH < want) continue;
            best_nch = 1; c.nw = KTf / CH; c.CH = CH; c.NCH = 1;
            break;
        }
    } else {     // (the split combine as 8 weight-streaming waves instead of 4 + helpers measured equal, profiles/r2h_*: not instantiated any more)
    // Fewer waves with longer K slices (log G5, round 6): once no workgroup barrier stands in front of the MFMAs (wave-local staging, log G2) a wave's
    // cost is its fixed part (row loads, LDS reduction leg, its place at the epilogue's barrier) more than its k-tiles — K = 768 as two waves of twelve
    // k-tiles instead of four of six (step graph 367.9 -> 363.6 us, headline +0.8 %), 1024 as four of eight instead of eight of four (medium.en
    // +2.9 %), 512 as two of eight (+0.3 %). Measured and left alone (profiles/r6as_*, r6ao_*): ONE wave of 24 k-tiles for K = 768 (-2.0 %: one wave
    // cannot keep 24 KiB of loads in flight AND the chain of 24 dependent MFMAs is 0.4 us), two waves of sixteen for K = 1024 (equal), K = 1280 as five
    // waves of eight (uneven over the four SIMDs: -1 %; it runs as four of ten, below), the row tiles of a batched step (12 windows per decode -1.8 %:
    // Mtot > 0 keeps the narrow slices), the split combine of the cross-attention output projection as three waves of eight / two of twelve with
    // three / four items peeled per lane (-1.5 % / -10 %: its waves are bound by the partials they gather, not by their count). Wide slices stay
    // within their launch bound (512 threads: an instantiation bound to 512 launched with 1024 is 'unspecified launch failure') and on SIMD-even counts.
    // WLX_G2_CHMAX (A/B builds): the widest chunk tried, 4..12 (6 = the pick until log G5). scripts/gemv_pick_probe.cpp prints the picks on the host.
    static const int chmax_env = [] { const char* e = wlx_ab("WLX_G2_CHMAX"); const int v = e ? atoi(e) : 12; return (v >= 4 && v <= 12) ? v : 12; }();
    // (8 / 12 k-tiles per wave: fp16 rows in only, staged rows (one row tile) — the split combine peels two items per lane for six k-tiles)
    // K = 1280 as four waves of ten k-tiles instead of eight of five (large-v3 step graph 1300 -> 1288 us, +1.1 %: profiles/r6au_*); WLX_G2_CH10=0
    // (A/B builds) = eight of five
    static const bool ch10 = [] { const char* e = wlx_ab("WLX_G2_CH10"); return !(e && e[0] == '0'); }();
    // Log G8 (round 6): the LayerNorm-fronted projections on PLAIN rows of one stream's step (cross-attention query where it is not fused, first MLP
    // projection) as four waves instead of eight for K = 1024 / 1280 — 8 / 10 k-tiles per wave, five rows in two row trips: large-v3 step graph
    // 1287 -> 1250 us (+3.5 %), medium.en 859 -> 832 us (+2.9 %), profiles/r6aw_*. WLX_G2_LN_WIDE=0 (A/B builds) = eight waves.
    static const bool ln_wide = [] { const char* e = wlx_ab("WLX_G2_LN_WIDE"); return !(e && e[0] == '0'); }();
    const bool ln_wide_here = ln_wide && p.in_mode == GEMV_IN_LN && p.M <= 8 && p.Mtot == 0 && (p.K == 1024 || p.K == 1280) && (p.xsrc == GEMV_X_PLAIN || xs_few_here);
    const int chmax = (p.in_mode == GEMV_IN_F16 && p.M <= 16 && p.Mtot == 0) ? chmax_env : ln_wide_here ? 10 : std::min(chmax_env, 6);
    for (int CH = chmax; CH >= 4; --CH) {
        if (CH != 12 && CH != 10 && CH != 8 && CH > 6) continue;
        if (CH == 10 && !ch10 && !ln_wide_here) continue;
        if (ln_wide_here && CH > 6 && CH * 128 != p.K) continue;
        if (KTf % CH) continue;
        const int q = KTf / CH;                     // = nw * NCH
        for (int nw = std::min(CH > 6 ? std::min(cap, 8) : cap, q); nw >= 1; --nw) {
            if (q % nw) continue;
            if (CH > 6 && nw > 4 && (nw & 3)) break;
            const int nch = q / nw;
            if (nch < best_nch) { best_nch = nch; c.nw = nw; c.CH = CH; c.NCH = nch; }
            break;
        }
    }
    }
    if (p.in_mode == GEMV_IN_LN && p.K == 384) {
        // d_model 384 (tiny / tiny.en, round 5): 12 k-tiles as six waves of two, so that six waves share the LayerNorm of the rows
        // (the general search above would pick two waves of six k-tiles: three LayerNorm trips for five rows)
        best_nch = 1; c.nw = 6; c.CH = 2; c.NCH = 1;
    }
    if (best_nch == (1 << 30)) return c;
    if (p.in_mode != GEMV_IN_F16 && c.NCH != 1) return c;
    c.LNV = 0;
    if (p.in_mode == GEMV_IN_LN) {
        if (p.K == 384) c.LNV = 15;                 // 1.5 x 256: see dec_gemv2_kernel
        else {
            if (p.K % 256 || p.K / 256 < 2 || p.K / 256 > 5) return c;
            c.LNV = p.K / 256;
        }
    }
    c.NTB = (p.out_mode == GEMV_OUT_F32 && p.N > 8192) ? 2 : 1;
    // more 16-column tiles than CUs (large-v3's first MLP projection: 320): two tiles per workgroup keep the launch to one
    // round of workgroups and halve the redundant LayerNorm prologues. WLX_GELU_NTB2=0 keeps one tile (A/B).
    if (p.in_mode == GEMV_IN_LN && p.out_mode == GEMV_OUT_GELU_F16 && p.xsrc == GEMV_X_PLAIN && (p.N + 15) / 16 > 256 && ((p.N + 15) / 16) % 2 == 0) c.NTB = 2;
    // row tiles of a batched step (round 4): every 16-column workgroup of a LayerNorm-fronted projection normalises its 16 rows again —
    // at 60 rows ~90 % of its instructions. Two column tiles per workgroup halve that redundant work (and the workgroup count) for the
    // wide projections (>= 128 tiles: QKV, first MLP projection). WLX_RT_NTB2=0 keeps one tile (A/B).
    if (p.Mtot > 0 && p.rt_nz > 0 && p.in_mode == GEMV_IN_LN && p.xsrc == GEMV_X_PLAIN &&
        (p.out_mode == GEMV_OUT_GELU_F16 || p.out_mode == GEMV_OUT_QKV) && (p.N + 15) / 16 >= 128 && ((p.N + 15) / 16) % 2 == 0) c.NTB = 2;
    // ... four where the tile count allows (60 rows, Whisper-small: first projection 6.6 -> 6.1 us, first MLP projection 6.5 -> 5.9 us; the 4 x 12
    // configuration +1.5 %, profiles/r4r_*). WLX_RT_NTB4=0 keeps two (A/B).
    if (c.NTB == 2 && p.Mtot > 0 && p.rt_nz > 0 && ((p.N + 15) / 16) % 4 == 0 && p.M <= 16) c.NTB = 4;
    // The row-tiled fp16-rows-in residual projections stage their 16 rows x K per 16-column workgroup as well. Two column tiles per
    // workgroup cost a single slot latency (4.7 -> 5.3 us per launch: half as many workgroups for a launch of 192) but save work, and with
    // three or more slots decoding on the device the GPU is work-bound: 4 x 12 windows +3 % (profiles/r4r_*). The engine
    // passes that situation in as GemvParams::busy_device. WLX_RT_F16_NTB2=0 / 1 forces it off / on (A/B).
    static const int rt_f16_ntb2 = [] { const char* e = wlx_ab("WLX_RT_F16_NTB2"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    const bool f16_wide = rt_f16_ntb2 >= 0 ? rt_f16_ntb2 == 1 : p.busy_device != 0;
    if (f16_wide && p.Mtot > 0 && p.rt_nz > 0 && p.M <= 16 && p.in_mode == GEMV_IN_F16 && p.out_mode == GEMV_OUT_RESID && p.xsrc == GEMV_X_PLAIN && ((p.N + 15) / 16) % 2 == 0) c.NTB = 2;   // (one row tile per chunk: the only two-tile instantiation)
    // (measured and dropped, profiles/r4t_*: two tiles for the N = d_model LayerNorm + query projection under a busy device — no change;
    // four tiles for the residual projections — spills at their 1024-thread launch bound, -17 %)
    c.MT = (p.M + 15) / 16;
    c.shm = sizeof(float) * (size_t)c.nw * c.NTB * c.MT * 256;
    const size_t xs_bytes = (size_t)p.M * (KTf * 32 + 8) * sizeof(half_t);    // fp16 activation rows
    // two tiles per workgroup are an optimisation, not a requirement: where their reduction buffer plus the staged rows pass
    // a CU's LDS (large-v3's first MLP projection at 41..48 rows: 49 + 124 KiB) one tile per workgroup still runs lean —
    // this shape used to fall back to the first-generation kernel (48-row prompt-prefill chunks of large-v3)
    if (c.NTB == 2 && p.out_mode == GEMV_OUT_GELU_F16 && c.shm + xs_bytes > WLX_G2_LDS_MAX) {
        c.NTB = 1;
        c.shm = sizeof(float) * (size_t)c.nw * c.NTB * c.MT * 256;
    }
    c.xstage = true;
    // (row tiles of a batched step may stage past the default 64 KiB — large-v3's K-split MLP projection: 16 x 2560 fp16 = 82 KiB)
    const size_t xstage_max = (p.Mtot > 0 && p.rt_nz > 0) ? (size_t)WLX_G2_LDS_MAX : (size_t)64 * 1024;
    if (p.in_mode == GEMV_IN_F16 && (c.MT > 1 || c.shm + xs_bytes > xstage_max)) c.xstage = false;   // fragments from global instead
    if (c.xstage) c.shm += xs_bytes;
    if (c.shm > WLX_G2_LDS_MAX) return c;                              // beyond a CU's LDS (160 KiB, less a margin): older kernel
    if (c.shm > 64 * 1024 && g_lds_optin_refused.load(std::memory_order_relaxed)) return c;   // the device refused the raised limit once
    c.ok = true;
    return c;
}

template <int CH, int LNV, int MT>
static bool gemv2_launch_ln(const GemvParams& p, const Gemv2Cfg& c, dim3 grid, dim3 block, hipStream_t s) {
#define WLX_G2(OUT_, NTB_, XS_) g2_launch<CH, LNV, GEMV_IN_LN, OUT_, NTB_, MT, XS_>(grid, block, c.shm, s, p)
    switch (p.out_mode) {
        case GEMV_OUT_QKV:
            if (c.NTB == 4 && MT == 1) { g2_launch<CH, LNV, GEMV_IN_LN, GEMV_OUT_QKV, 4, 1, GEMV_X_PLAIN>(grid, block, c.shm, s, p); return true; }
            if (c.NTB == 2) WLX_G2(GEMV_OUT_QKV, 2, GEMV_X_PLAIN);
            else WLX_G2(GEMV_OUT_QKV, 1, GEMV_X_PLAIN);
            return true;
        case GEMV_OUT_F16:
            WLX_G2(GEMV_OUT_F16, 1, GEMV_X_PLAIN); return true;
        case GEMV_OUT_GELU_F16:
            if (c.NTB == 4 && MT == 1) { g2_launch<CH, LNV, GEMV_IN_LN, GEMV_OUT_GELU_F16, 4, 1, GEMV_X_PLAIN>(grid, block, c.shm, s, p); return true; }
            if (c.NTB == 2) WLX_G2(GEMV_OUT_GELU_F16, 2, GEMV_X_PLAIN);
            else WLX_G2(GEMV_OUT_GELU_F16, 1, GEMV_X_PLAIN);
            return true;
        case GEMV_OUT_F32:
            if (c.NTB == 2) WLX_G2(GEMV_OUT_F32, 2, GEMV_X_PLAIN);
            else WLX_G2(GEMV_OUT_F32, 1, GEMV_X_PLAIN);
            return true;
        default: return false;
    }
#undef WLX_G2
}
// the first projection of a layer reading slab / embedding rows (one row tile): its own (CH, LNV) pairs, see gemv2_cfg
template <int CH, int LNV, int MT>
static bool gemv2_launch_qkv_xs_mt(const GemvParams& p, const Gemv2Cfg& c, dim3 grid, dim3 block, hipStream_t s) {
    if (p.xsrc == GEMV_X_SLABS) g2_launch<CH, LNV, GEMV_IN_LN, GEMV_OUT_QKV, 1, MT, GEMV_X_SLABS>(grid, block, c.shm, s, p);
    else g2_launch<CH, LNV, GEMV_IN_LN, GEMV_OUT_QKV, 1, MT, GEMV_X_EMBED>(grid, block, c.shm, s, p);
    return true;
}
template <int CH, int LNV>
static bool gemv2_launch_qkv_xs(const GemvParams& p, const Gemv2Cfg& c, dim3 grid, dim3 block, hipStream_t s) {
    return c.MT == 1 ? gemv2_launch_qkv_xs_mt<CH, LNV, 1>(p, c, grid, block, s)
         : c.MT == 2 ? gemv2_launch_qkv_xs_mt<CH, LNV, 2>(p, c, grid, block, s)
                     : gemv2_launch_qkv_xs_mt<CH, LNV, 3>(p, c, grid, block, s);
}
template <int CH, int MT>
static bool gemv2_launch_other(const GemvParams& p, const Gemv2Cfg& c, dim3 grid, dim3 block, hipStream_t s) {
    if (p.in_mode == GEMV_IN_F16) {
        if (p.out_mode == GEMV_OUT_SLAB) { g2_launch<CH, 1, GEMV_IN_F16, GEMV_OUT_SLAB, 1, MT, GEMV_X_PLAIN>(grid, block, c.shm, s, p); return true; }
        if (p.xsrc == GEMV_X_SLABS) { g2_launch<CH, 1, GEMV_IN_F16, GEMV_OUT_RESID, 1, MT, GEMV_X_SLABS>(grid, block, c.shm, s, p); return true; }
        if (c.NTB == 2 && MT == 1) { g2_launch<CH, 1, GEMV_IN_F16, GEMV_OUT_RESID, 2, 1, GEMV_X_PLAIN>(grid, block, c.shm, s, p); return true; }
        g2_launch<CH, 1, GEMV_IN_F16, GEMV_OUT_RESID, 1, MT, GEMV_X_PLAIN>(grid, block, c.shm, s, p);
    } else g2_launch<CH, 1, GEMV_IN_XATTN, GEMV_OUT_RESID, 1, MT, GEMV_X_PLAIN>(grid, block, c.shm, s, p);
    return true;
}
// the (CH, LNV) pairs of the Whisper family: d_model 512 (4,2), 768 (6,3), 1024 (4,4), 1280 (5,5)
static bool gemv2_launch(const GemvParams& p0, const Gemv2Cfg& c, hipStream_t s) {
    GemvParams p = p0;
    p.KTW = c.CH * c.NCH; p.NCH = c.NCH; p.xstage = c.xstage ? 1 : 0; p.nwm = c.nw;
#ifdef WLX_TRACE
    { static thread_local char nm[512][48]; const int q = g_trace_seq < 512 ? g_trace_seq : 511;
      snprintf(nm[q], 48, "gemv2<%d,%d,%d> N%d K%d", p.in_mode, p.out_mode, p.xsrc, p.N, p.K); p.trc = trace_next(nm[q]); }
#endif
    const int NT_total = (p.N + 15) / 16;
    if (p.Mtot > 0 && p.chunk <= 0) p.chunk = 48;
    dim3 grid((NT_total + c.NTB - 1) / c.NTB, p.out_mode == GEMV_OUT_SLAB ? p.KT / p.KTS : 1, p.Mtot > 0 ? (p.Mtot + p.chunk - 1) / p.chunk : 1), block(c.nw * 64);
    if (p.Mtot > 0 && p.rt_nz > 0) {       // row tiles folded into x (see dec_gemv2_kernel)
        p.rt_tiles = (int)grid.x;
        p.rt_magic = 65536 / p.rt_nz + 1;
        grid.x = ((grid.x + 7) / 8) * 8 * p.rt_nz;
        grid.z = 1;
    }
    if (p.in_mode == GEMV_IN_XATTN && !(WLX_XCOMB_WAVE != 0 && c.MT == 1)) {      // helper waves for the combine: one thread per (row, head, 4-float group), <= 1024
        const int want = (p.M * p.H * 8 + 63) / 64;
        block.x = 64 * std::max(c.nw, std::min(c.MT > 1 ? 8 : 16, want));
    }
#define WLX_G2_LN(CH_, LNV_) (c.MT == 1 ? gemv2_launch_ln<CH_, LNV_, 1>(p, c, grid, block, s) : c.MT == 2 ? gemv2_launch_ln<CH_, LNV_, 2>(p, c, grid, block, s) : gemv2_launch_ln<CH_, LNV_, 3>(p, c, grid, block, s))
#define WLX_G2_OT(CH_) (c.MT == 1 ? gemv2_launch_other<CH_, 1>(p, c, grid, block, s) : c.MT == 2 ? gemv2_launch_other<CH_, 2>(p, c, grid, block, s) : gemv2_launch_other<CH_, 3>(p, c, grid, block, s))
    if (p.in_mode == GEMV_IN_LN && p.xsrc != GEMV_X_PLAIN) {
        if (c.CH == 4 && c.LNV == 3) return gemv2_launch_qkv_xs<4, 3>(p, c, grid, block, s);
        if (c.CH == 3 && c.LNV == 3) return gemv2_launch_qkv_xs<3, 3>(p, c, grid, block, s);
        if (c.CH == 2 && c.LNV == 2) return gemv2_launch_qkv_xs<2, 2>(p, c, grid, block, s);
        if (c.CH == 2 && c.LNV == 15) return c.MT == 1 ? gemv2_launch_qkv_xs_mt<2, 15, 1>(p, c, grid, block, s) : false;
        if (c.CH == 4 && c.LNV == 4) return gemv2_launch_qkv_xs<4, 4>(p, c, grid, block, s);
        if (c.CH == 5 && c.LNV == 5) return gemv2_launch_qkv_xs<5, 5>(p, c, grid, block, s);
        if (c.CH == 6 && c.LNV == 3) return c.MT == 1 ? gemv2_launch_qkv_xs_mt<6, 3, 1>(p, c, grid, block, s) : false;
        if (c.CH == 8 && c.LNV == 4) return c.MT == 1 ? gemv2_launch_qkv_xs_mt<8, 4, 1>(p, c, grid, block, s) : false;
        if (c.CH == 10 && c.LNV == 5) return c.MT == 1 ? gemv2_launch_qkv_xs_mt<10, 5, 1>(p, c, grid, block, s) : false;
        return false;
    }
    if (p.in_mode == GEMV_IN_LN) {
        if (c.CH == 6 && c.LNV == 3) return WLX_G2_LN(6, 3);
        if (c.CH == 5 && c.LNV == 5) return WLX_G2_LN(5, 5);
        if (c.CH == 10 && c.LNV == 5) return c.MT == 1 ? gemv2_launch_ln<10, 5, 1>(p, c, grid, block, s) : false;
        if (c.CH == 8 && c.LNV == 4) return c.MT == 1 ? gemv2_launch_ln<8, 4, 1>(p, c, grid, block, s) : false;
        if (c.CH == 4 && c.LNV == 2) return WLX_G2_LN(4, 2);
        if (c.CH == 4 && c.LNV == 4) return WLX_G2_LN(4, 4);
        if (c.CH == 2 && c.LNV == 15) return c.MT == 1 ? gemv2_launch_ln<2, 15, 1>(p, c, grid, block, s) : false;   // (one row tile: batched rows run as row tiles)
        return false;
    }
    switch (c.CH) {
        case 12: return c.MT == 1 && p.in_mode == GEMV_IN_F16 ? gemv2_launch_other<12, 1>(p, c, grid, block, s) : false;
        case 10: return c.MT == 1 && p.in_mode == GEMV_IN_F16 ? gemv2_launch_other<10, 1>(p, c, grid, block, s) : false;
        case 8: return c.MT == 1 && p.in_mode == GEMV_IN_F16 ? gemv2_launch_other<8, 1>(p, c, grid, block, s) : false;
        case 6: return WLX_G2_OT(6);
        case 5: return WLX_G2_OT(5);
        default: return WLX_G2_OT(4);
    }
#undef WLX_G2_LN
#undef WLX_G2_OT
}
static bool gemv2_ok(const GemvParams& p, Gemv2Cfg* out) {
    const Gemv2Cfg c = gemv2_cfg(p);
    if (!c.ok) return false;
    if (p.in_mode == GEMV_IN_LN && p.xsrc != GEMV_X_PLAIN) {
        const bool pair = (c.CH == 4 && c.LNV == 3) || (c.CH == 3 && c.LNV == 3) || (c.CH == 2 && c.LNV == 2) || (c.CH == 4 && c.LNV == 4) || (c.CH == 5 && c.LNV == 5) ||
                          (c.MT == 1 && ((c.CH == 6 && c.LNV == 3) || (c.CH == 8 && c.LNV == 4) || (c.CH == 10 && c.LNV == 5))) ||
                          (c.CH == 2 && c.LNV == 15 && c.MT == 1);
        if (!pair) return false;
    } else if (p.in_mode == GEMV_IN_LN) {
        const bool pair = (c.CH == 6 && c.LNV == 3) || (c.CH == 5 && c.LNV == 5) || (c.CH == 4 && c.LNV == 2) || (c.CH == 4 && c.LNV == 4) ||
                          (c.CH == 10 && c.LNV == 5 && c.MT == 1) || (c.CH == 8 && c.LNV == 4 && c.MT == 1) ||
                          (c.CH == 2 && c.LNV == 15 && c.MT == 1);
        if (!pair) return false;
    }
    if (out) *out = c;
    return true;
}
// more than 48 rows (prompt prefill): the lean kernel in row chunks of 48 (grid.z), configured for a full chunk
// 17..WLX_ROWTILE_MAX rows (batched decode steps): row chunks of one 16-row tile folded into blockIdx.x (round 4, see
// dec_gemv2_kernel). WLX_ROWTILE=0 restores the 48-row form (A/B).
static GemvParams gemv_chunked(const GemvParams& p) {
    static const bool rt_on = [] { const char* e = wlx_ab("WLX_ROWTILE"); return !(e && e[0] == '0'); }();
    constexpr int rt_max = WLX_MAX_DEC_ROWS;
    // rows per tile: 16 (one MFMA row tile per workgroup; 32- and 48-row tiles were measured in round 4 for <= 64 rows and lost). WLX_ROWTILE_CHUNK
    // (A/B builds) = 32 / 48: two / three row tiles per workgroup share one pass over the weight tile
    static const int rt_chunk_env = [] { const char* e = wlx_ab("WLX_ROWTILE_CHUNK"); const int v = e ? atoi(e) : 0; return (v == 16 || v == 32 || v == 48) ? v : 0; }();
    int rt_chunk = 16;
    // Round 5 (wide batches): the fp16-rows-in residual projections (attention output, cross-attention output, MLP down: 1024-thread
    // workgroups, one per CU, no LayerNorm prologue to repeat) take two or three row tiles per workgroup where that saves ROUNDS of
    // workgroups: N = 768 at 120 rows is 48 x 8 = 384 workgroups on 256 CUs with 16-row tiles, 192 with 32-row tiles (6.7 -> 5.8 us per
    // launch; large-v3 at 160 rows 16.7 -> 13.9 us; profiles/r5h_rowtile_chunk_by_rows.txt). Cost model: rounds x (1 + 0.35 per extra row
    // tile). The LayerNorm-fronted projections stay on 16 rows (every workgroup re-normalises its rows: 32-row tiles measured 10-40 % slower).
    if (p.in_mode == GEMV_IN_F16 && p.out_mode == GEMV_OUT_RESID && p.xsrc == GEMV_X_PLAIN && !p.busy_device && p.M > 64) {
        static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount; return n; }();
        const int tiles = (p.N + 15) / 16;
        double best = 1e300;
        for (int mt = 1; mt <= 3; ++mt) {
            const long wgs = (long)tiles * ((p.M + 16 * mt - 1) / (16 * mt));
            const double cost = (double)((wgs + n_cu - 1) / n_cu) * (1.0 + 0.35 * (mt - 1));
            if (cost < best - 1e-9) { best = cost; rt_chunk = 16 * mt; }
        }
    }
    if (rt_chunk_env) rt_chunk = rt_chunk_env;
    GemvParams q = p;
    if (p.Mtot != 0 || p.in_mode == GEMV_IN_XATTN) return q;
    // Which projections: measured per kernel at 20 / 40 / 60 rows (profiles/r4a-c_*): row tiles win wherever the launch has few
    // column tiles (N = d_model: 7.1 -> 4.2 us at 60 rows, large-v3 6.7 -> 4.8 us at 40) and for Whisper-small's wide ones
    // (first MLP projection 10.3 -> 6.9 us); large-v3's N = 3 d / 4 d projections (10-13 MB of weights re-read per row tile
    // through L2) were faster as three-tile workgroups (8.7 vs 11.8 us, 9.1 vs 9.8 us) — with ONE column tile per workgroup. With the four
    // column tiles per workgroup the row-tiled LayerNorm projections got later in the round they are not (large-v3, 40 rows: first
    // projection 8.8 -> 7.7 us, first MLP projection 9.2 -> 7.6 us, step 2183 -> 2087 us; profiles/r4lv3rt_decode_step.txt): every
    // LayerNorm-fronted projection whose tile count divides by four is cut too from three row tiles up (60 rows: 3157 -> 2857 us; 20 rows,
    // two row tiles: 1845 -> 1905 us, left as it was). WLX_ROWTILE_WIDE=0 = the earlier policy (A/B).
    const bool ln_wide4 = rt_chunk == 16 && p.M > 32 && p.in_mode == GEMV_IN_LN && p.xsrc == GEMV_X_PLAIN && (p.out_mode == GEMV_OUT_QKV || p.out_mode == GEMV_OUT_GELU_F16) &&
                          (p.N & 15) == 0 && ((p.N >> 4) & 3) == 0;
    const bool rt_shape = p.N <= 1536 || (long)p.N * p.K <= 3200000L || ln_wide4;
    if (rt_on && !g_decode_v1 && p.M > rt_chunk && p.M <= rt_max && rt_shape) { q.Mtot = p.M; q.M = rt_chunk; q.chunk = rt_chunk; q.rt_nz = (p.M + rt_chunk - 1) / rt_chunk; }
    else if (p.M > 48 && p.xsrc != GEMV_X_EMBED) { q.Mtot = p.M; q.M = 48; q.chunk = 48; }
    return q;
}
// (Round 6, measured and dropped — "wide passes": from 64 / 128 rows the LayerNorms as their own launch (fp16 rows, the prologue's arithmetic) and
// every K = d_model projection with fp16 rows in on 64-ROW tiles (MT = 4, one or two column tiles per workgroup), so that a weight tile is fetched
// from L2 once per 64 rows instead of once per 16. Parity green (210 GPU tests). Whisper-small at 120 / 240 rows: step 1.007 / 1.442 ms against
// 0.913 / 1.365 ms on the 16-row tiles; large-v3 at 80 / 160 rows: 3.61 / 4.85 against 3.06 / 5.03 ms — the three LayerNorm launches per layer
// (2 us each) and the 64-row workgroups' lower occupancy cost what the saved L2 re-reads give back; only large-v3 at 160 rows gains (3.6 %).
// profiles/r6c_wide_rows_*, r6d_wide_rows_*; DESIGN.md §7.3 B4. The 16-row tiles stay.)
static bool vocab2_ok(const GemvParams& p);   // (dec_vocab_kernel, below)
bool dec_gemv_is_lean(const GemvParams& p) { return vocab2_ok(p) || gemv2_ok(gemv_chunked(p), nullptr); }

int dec_gemv_slab_split(int M, int K, int N) {
    if (WLX_FC2_KS < 2) return 0;                                             // (the slab count is a compile-time constant of the consumers: -DWLX_FC2_KS)
    if (g_decode_v1 || M < 1 || K % 32 || (K / 32) % WLX_FC2_KS || K < 2048) return 0;
    GemvParams p0{};
    p0.in_mode = GEMV_IN_F16; p0.out_mode = GEMV_OUT_SLAB; p0.M = M; p0.K = K; p0.KT = K / 32; p0.N = N; p0.KTS = p0.KT / WLX_FC2_KS;
    static const float dummy_bias = 0.f;
    p0.bias = &dummy_bias;                                                    // (cfg only asks whether there is one)
    const GemvParams p = gemv_chunked(p0);                                    // row chunks: decided for a full chunk (16 or 48 rows)
    Gemv2Cfg c;
    if (!gemv2_ok(p, &c) || (p.M <= 16 && !c.xstage)) return 0;
    return WLX_FC2_KS;
}

static bool vocab2_ok(const GemvParams& p);   // (dec_vocab_kernel, below)
static int vocab2_chunk_rows(int K);
const char* dec_gemv_kernel_name(const GemvParams& p_any) {
    static thread_local char buf[64];
    if (vocab2_ok(p_any)) {
        snprintf(buf, sizeof(buf), "dec_vocab_kernel<%d, %d, %d>", p_any.KT, p_any.KT == 24 ? 6 : p_any.KT == 40 ? 5 : p_any.KT == 12 ? 3 : 4, (std::min(p_any.M, vocab2_chunk_rows(p_any.K)) + 15) / 16);
        return buf;
    }
    const GemvParams p = gemv_chunked(p_any);
    const int MT = (p.M + 15) / 16;
    Gemv2Cfg c2;
    if (gemv2_ok(p, &c2)) {
        snprintf(buf, sizeof(buf), "dec_gemv2_kernel<%d, %d, %d, %d, %d, %d, %d>", c2.CH, c2.LNV ? c2.LNV : 1, p.in_mode, p.out_mode, c2.NTB, c2.MT, p.xsrc);
        return buf;
    }
    snprintf(buf, sizeof(buf), "dec_gemv_kernel<%d, %d, %d>", MT > 4 ? 4 : MT, MT == 1 ? 2 : 1, p.in_mode);
    return buf;
}

// ------------------------------------------------------------------ vocabulary projection: final LayerNorm + tied output projection
// (round 4) The one bandwidth-sized launch of a decode step: V x d_model fp16 (80 MB Whisper-small, 133 MB large-v3) against <= 64
// rows. As an instance of dec_gemv2_kernel it was 1621 workgroups that each normalised ALL rows before their 48 KiB of weights
// could be used — 19 us at 5 rows (4.2 TB/s) but 51-80 us at 60 rows (the LayerNorm prologue, not HBM: 1621 x 60 rows x 3 KiB of
// fp32 loads, ~11 ns per wave-level load per CU). Here a workgroup is 8 waves that share ONE LayerNorm of the rows (fp16 rows in
// LDS) and then each wave owns a PAIR of 16-column tiles over the whole K: every weight fragment is loaded once (non-temporal,
// straight into registers, two chunks of KC k-tiles x 2 tiles in flight = 24 KiB per wave) and multiplied against all MT row
// tiles from LDS — no K split, no cross-wave reduction, fp32 logits leave as 16-byte pieces. 203 workgroups for V = 51864: one
// round on 256 CUs. A row's result does not depend on how many rows share the launch (same code, same summation order).
struct VocabParams {
    const float* X; long ldx; const float* gamma; const float* beta;
    const half_t* Wp; int M, N, NT;            // rows, real outputs, 16-column tiles of the packed weights
    float* Y; long ldy;
    const float* slab; long slab_stride;       // SLABS: the rows are X + the WLX_FC2_KS partial-sum slabs of the LAST layer's K-split MLP projection
    WLX_TR_FIELD
};
// SLABS (round 5, one row tile): the last decoder layer's MLP output projection used to stay a single launch because its consumer — this
// projection, then 1621 workgroups that would each have summed the slabs — made the split a loss; as ONE launch with K = 4 d_model it is
// the slowest projection of the step (11-13 us at 5 rows against 4.3 us for the K-split form, profiles/r5b_*). With 203 workgroups that
// share one LayerNorm the slab reads are three small loads per row, so the last layer splits like the others.
template <int KT, int KC, int MT, bool SLABS>
__global__ __launch_bounds__(512) void dec_vocab_kernel(VocabParams p) {
    static_assert(!SLABS || MT == 1, "slab rows: decode steps of one row tile");
    constexpr int K = KT * 32, LNV = (K + 255) / 256, NC = KT / KC, LDXS = K + 8;
    constexpr bool LNT = (K % 256) != 0;                    // d_model 384 = 1.5 x 256: the second float4 unit is live on lanes 0..31 only (see dec_gemv2_kernel)
    static_assert((K % 256 == 0 || K == 384) && KT % KC == 0 && NC % 2 == 0, "d_model a multiple of 256 (or 384); an even number of K chunks");
    constexpr int RPT = (MT == 1) ? 2 : 4;                  // LayerNorm rows a wave requests per trip (8 waves: 16 / 32 rows per trip)
    constexpr int NTRIP = (MT * 16 + 8 * RPT - 1) / (8 * RPT);
    extern __shared__ __attribute__((aligned(16))) half_t vxs[];   // [M][LDXS] fp16 LayerNorm rows
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WLX_TR_BEGIN();
    const int pair = blockIdx.x * 8 + wave;
    const int t0 = (2 * pair < p.NT) ? 2 * pair : p.NT - 1, t1 = (2 * pair + 1 < p.NT) ? 2 * pair + 1 : p.NT - 1;
    const bool tail_on = !LNT || lane < 32;
    const int tback = LNT ? (tail_on ? 0 : lane) : 0;
    // ---- first trip's rows FIRST (vmcnt retires in order: the LayerNorm must not wait behind the weight stream)
    float4 x[RPT][LNV];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = (wave + 8 * i < p.M) ? wave + 8 * i : p.M - 1;
        const float4* x4 = reinterpret_cast<const float4*>(p.X + (long)r * p.ldx) + lane;
#pragma unroll
        for (int j = 0; j < LNV; ++j) x[i][j] = x4[64 * j - ((j == LNV - 1) ? tback : 0)];
    }
    float4 xsl[SLABS ? WLX_FC2_KS : 1][RPT][LNV];
    if constexpr (SLABS) {
#pragma unroll
        for (int q = 0; q < WLX_FC2_KS; ++q)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int r = (wave + 8 * i < p.M) ? wave + 8 * i : p.M - 1;
                const float4* s4 = reinterpret_cast<const float4*>(p.slab + q * p.slab_stride + (long)r * p.ldx) + lane;
#pragma unroll
                for (int j = 0; j < LNV; ++j) xsl[q][i][j] = s4[64 * j - ((j == LNV - 1) ? tback : 0)];
            }
    }
    float4 gq[LNV], bq[LNV];
    {
        const float4* g4 = reinterpret_cast<const float4*>(p.gamma) + lane;
        const float4* b4 = reinterpret_cast<const float4*>(p.beta) + lane;
#pragma unroll
        for (int j = 0; j < LNV; ++j) { const int tb = (j == LNV - 1) ? tback : 0; gq[j] = g4[64 * j - tb]; bq[j] = b4[64 * j - tb]; }
    }
    asm volatile("" ::: "memory");                         // compile-time fence: the weight requests stay behind the row requests
    const half_t* wq[2] = {p.Wp + (long)t0 * KT * 512 + lane * 8, p.Wp + (long)t1 * KT * 512 + lane * 8};
    f16x8 wf[2][KC][2];                                     // [ring buffer][k-tile of the chunk][tile of the pair]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < KC; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[b][j][i] = ld_nt_f16x8(wq[i] + (b * KC + j) * 512);
    constexpr float invK = 1.0f / (float)K;
    auto ln_row = [&](float4 (&xr)[LNV], int r, bool keep) {
        if constexpr (LNT) { if (!tail_on) xr[LNV - 1] = make_float4(0.f, 0.f, 0.f, 0.f); }
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < LNV; ++j) sm += (xr[j].x + xr[j].y) + (xr[j].z + xr[j].w);
        const float mean = dpp_wave_sum(sm) * invK;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LNV; ++j) {
            xr[j].x -= mean; xr[j].y -= mean; xr[j].z -= mean; xr[j].w -= mean;
            if constexpr (LNT) { if (j == LNV - 1 && !tail_on) xr[j] = make_float4(0.f, 0.f, 0.f, 0.f); }
            q += (xr[j].x * xr[j].x + xr[j].y * xr[j].y) + (xr[j].z * xr[j].z + xr[j].w * xr[j].w);
        }
        const float rstd = rsqrtf(dpp_wave_sum(q) * invK + 1e-5f);
        half_t* dst = vxs + (long)r * LDXS + lane * 4;
#pragma unroll
        for (int j = 0; j < LNV; ++j) {
            const f16x4 hv = {(half_t)(xr[j].x * rstd * gq[j].x + bq[j].x), (half_t)(xr[j].y * rstd * gq[j].y + bq[j].y),
                              (half_t)(xr[j].z * rstd * gq[j].z + bq[j].z), (half_t)(xr[j].w * rstd * gq[j].w + bq[j].w)};
            if (keep && (j < LNV - 1 || tail_on)) *reinterpret_cast<f16x4*>(dst + 256 * j) = hv;
        }
    };
    // first trip: straight-line and unconditional (a wave without a row normalises the clamped row it loaded and keeps nothing)
    if constexpr (SLABS) {                                  // the row = ((x + s0) + s1): the association of every other consumer of the slabs
#pragma unroll
        for (int q = 0; q < WLX_FC2_KS; ++q)
#pragma unroll
            for (int i = 0; i < RPT; ++i)
#pragma unroll
                for (int j = 0; j < LNV; ++j) { x[i][j].x += xsl[q][i][j].x; x[i][j].y += xsl[q][i][j].y; x[i][j].z += xsl[q][i][j].z; x[i][j].w += xsl[q][i][j].w; }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) ln_row(x[i], (wave + 8 * i < p.M) ? wave + 8 * i : p.M - 1, wave + 8 * i < p.M);
#pragma unroll 1
    for (int tr = 1; tr < NTRIP; ++tr) {                    // (49..64 rows, or 33..48: a second trip behind the weight stream)
        const int rb = wave + 8 * RPT * tr;
        if (rb >= p.M) break;
        float4 y[RPT][LNV];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = (rb + 8 * i < p.M) ? rb + 8 * i : p.M - 1;
            const float4* x4 = reinterpret_cast<const float4*>(p.X + (long)r * p.ldx) + lane;
#pragma unroll
            for (int j = 0; j < LNV; ++j) y[i][j] = x4[64 * j - ((j == LNV - 1) ? tback : 0)];
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) ln_row(y[i], (rb + 8 * i < p.M) ? rb + 8 * i : p.M - 1, rb + 8 * i < p.M);
    }
    WLX_TR_MARK(1);
    __syncthreads();
    // ---- the pair's columns over the whole K: chunk ch from ring buffer ch & 1, refilled with chunk ch + 2 behind its MFMAs
    const half_t* xr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xr[mt] = vxs + (long)((mt * 16 + c < p.M) ? mt * 16 + c : p.M - 1) * LDXS + g * 8;   // rows >= M re-read a valid row (never stored)
    f32x4 acc[2][MT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[i][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NC; ++ch) {
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            f16x8 xf[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const f16x8*>(xr[mt] + (ch * KC + j) * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = mfma16(wf[ch & 1][j][i], xf[mt], acc[i][mt]);
        }
        if (ch + 2 < NC) {
#pragma unroll
            for (int j = 0; j < KC; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[ch & 1][j][i] = ld_nt_f16x8(wq[i] + ((ch + 2) * KC + j) * 512);
        }
    }
    WLX_TR_MARK(2);
    // ---- fp32 logits: lane (c, g) holds columns g*4 .. g*4+3 of row mt*16 + c
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int nt = 2 * pair + i;
        if (nt >= p.NT) continue;
        const int n = nt * 16 + g * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + c;
            if (row >= p.M) continue;
            float* yp = p.Y + (long)row * p.ldy + n;
            const f32x4 v = acc[i][mt];
            if (n + 3 < p.N) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            else { if (n < p.N) yp[0] = v[0]; if (n + 1 < p.N) yp[1] = v[1]; if (n + 2 < p.N) yp[2] = v[2]; }
        }
    }
    WLX_TR_END(p.trc);
}

template <int KT, int KC, int MT, bool SLABS = false>
static void vocab_go(const VocabParams& p, hipStream_t s) {
    const size_t shm = (size_t)p.M * (KT * 32 + 8) * sizeof(half_t);
    if (shm > 64 * 1024) {          // > 64 KiB of dynamic LDS: opt in once per device (first launch of a shape happens outside capture)
        static std::atomic<signed char> granted[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && granted[dev].load(std::memory_order_acquire) == 0) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_vocab_kernel<KT, KC, MT, SLABS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    WLX_G2_LDS_MAX) == hipSuccess) granted[dev].store(1, std::memory_order_release);
            else {                  // this launch fails with the runtime's own error (reported for THIS call); later calls take the general kernel (vocab2_ok)
                (void)hipGetLastError();
                g_lds_optin_refused.store(true);
                fprintf(stderr, "[wlx] device %d refused %d KiB of dynamic LDS: the batched vocabulary projection falls back to the general kernel\n", dev, WLX_G2_LDS_MAX / 1024);
            }
        }
    }
    const int pairs = (p.NT + 1) / 2;
    hipLaunchKernelGGL((dec_vocab_kernel<KT, KC, MT, SLABS>), dim3((pairs + 7) / 8), dim3(512), shm, s, p);
}
template <int KT, int KC>
static void vocab_go_mt(const VocabParams& p, hipStream_t s) {
    if (p.slab) { vocab_go<KT, KC, 1, true>(p, s); return; }        // (vocab2_ok: one row tile)
    switch ((p.M + 15) / 16) {
        case 1: vocab_go<KT, KC, 1>(p, s); break;
        case 2: vocab_go<KT, KC, 2>(p, s); break;
        case 3: vocab_go<KT, KC, 3>(p, s); break;
        default: vocab_go<KT, KC, 4>(p, s); break;
    }
}
// rows one launch of dec_vocab_kernel takes: its fp16 LayerNorm rows must fit the workgroup's LDS (152 KiB: d_model <= 1024 64 rows,
// large-v3 60 -> 48 = three whole row tiles). A wider pass (round 5: up to WLX_MAX_DEC_ROWS rows per step) runs as consecutive row
// chunks, each streaming the weights again (Whisper-small 80 MB = ~25 us per 64 rows of a ~1 ms step).
static int vocab2_chunk_rows(int K) {
    int r = 64;
    while (r > 16 && (size_t)r * (K + 8) * sizeof(half_t) > WLX_G2_LDS_MAX) r -= 16;
    return r;
}
static bool vocab2_ok(const GemvParams& p) {
    if (g_decode_v1 || p.in_mode != GEMV_IN_LN || p.out_mode != GEMV_OUT_F32 || p.bias || p.Mtot != 0) return false;
    if (p.xsrc != GEMV_X_PLAIN && !(p.xsrc == GEMV_X_SLABS && p.M <= 16 && p.slab != nullptr)) return false;   // rows + slabs: decode steps of one row tile
    if (p.M < 1 || p.M > WLX_MAX_DEC_ROWS || p.K != p.KT * 32 || p.N < 256) return false;
    if (!(p.KT == 12 || p.KT == 16 || p.KT == 24 || p.KT == 32 || p.KT == 40)) return false;
    const int rows = std::min(p.M, vocab2_chunk_rows(p.K));
    const size_t shm = (size_t)rows * (p.K + 8) * sizeof(half_t);
    if (shm > 64 * 1024 && g_lds_optin_refused.load(std::memory_order_relaxed)) return false;   // the device refused the raised LDS limit once: general kernel
    return shm <= WLX_G2_LDS_MAX;
}
static void vocab2_launch(const GemvParams& g, hipStream_t s) {
    const int CH = vocab2_chunk_rows(g.K);
    for (int r0 = 0; r0 < g.M; r0 += CH) {
        VocabParams p{};
        p.X = g.X + (long)r0 * g.ldx; p.ldx = g.ldx; p.gamma = g.gamma; p.beta = g.beta; p.Wp = g.Wp; p.M = std::min(CH, g.M - r0); p.N = g.N; p.NT = (g.N + 15) / 16;
        p.Y = g.Y + (long)r0 * g.ldy; p.ldy = g.ldy;
        if (g.xsrc == GEMV_X_SLABS) { p.slab = g.slab; p.slab_stride = g.slab_stride; }
#ifdef WLX_TRACE
        p.trc = trace_next("vocab2");
#endif
        switch (g.KT) {
            case 12: vocab_go_mt<12, 3>(p, s); break;
            case 16: vocab_go_mt<16, 4>(p, s); break;
            case 24: vocab_go_mt<24, 6>(p, s); break;
            case 32: vocab_go_mt<32, 4>(p, s); break;
            default: vocab_go_mt<40, 5>(p, s); break;
        }
    }
}

template <int MT, int NTB>
static void gemv_dispatch_in(const GemvParams& p, dim3 grid, dim3 block, size_t shm, hipStream_t s) {
    switch (p.in_mode) {
        case GEMV_IN_LN: hipLaunchKernelGGL((dec_gemv_kernel<MT, NTB, GEMV_IN_LN>), grid, block, shm, s, p); break;
        case GEMV_IN_F16: hipLaunchKernelGGL((dec_gemv_kernel<MT, NTB, GEMV_IN_F16>), grid, block, shm, s, p); break;
        default: hipLaunchKernelGGL((dec_gemv_kernel<MT, NTB, GEMV_IN_XATTN>), grid, block, shm, s, p); break;
    }
}

void launch_dec_gemv(const GemvParams& p_any, hipStream_t s) {
    if (vocab2_ok(p_any)) { vocab2_launch(p_any, s); return; }
    Gemv2Cfg c2;
    const GemvParams pc = gemv_chunked(p_any);
    if (gemv2_ok(pc, &c2) && gemv2_launch(pc, c2, s)) return;
    if (p_any.M > 16 * WLX_MAX_MT) {
        // the general kernel holds WLX_MAX_MT row tiles per launch: a wider pass (round 5) runs as consecutive row chunks on rebased row
        // pointers (64 rows; the split-combine prologue indexes its partials by (item, row in item), so its chunks are whole items)
        const int CHK = (p_any.in_mode == GEMV_IN_XATTN && p_any.R > 0) ? (16 * WLX_MAX_MT / p_any.R) * p_any.R : 16 * WLX_MAX_MT;
        for (int r0 = 0; r0 < p_any.M; r0 += CHK) {
            GemvParams q = p_any;
            q.M = std::min(CHK, p_any.M - r0);
            if (q.X) q.X += (long)r0 * q.ldx;
            if (q.Xh) q.Xh += (long)r0 * q.ldxh;
            if (q.Yh) q.Yh += (long)r0 * q.ldyh;
            if (q.Y) q.Y += (long)r0 * q.ldy;
            if (q.Xres) q.Xres += (long)r0 * q.ldxres;
            if (q.row_cache) q.row_cache += r0;
            if (q.row_pos) q.row_pos += r0;
            if (q.in_mode == GEMV_IN_XATTN) {
                const int items0 = r0 / q.R;
                q.part_o += (long)items0 * q.H * WLX_XSPLIT * 16 * 64;
                q.part_ml += (long)items0 * q.H * 16 * WLX_XSPLIT * 2;
            }
            launch_dec_gemv(q, s);
        }
        return;
    }
    const GemvParams& p0 = p_any;
    const GemvParams& p = p0;
    const int MT = (p.M + 15) / 16;
    const int NT_total = (p.N + 15) / 16;
    // waves per workgroup: enough K-split that each wave streams <= GV_CH k-tiles per chunk and,
    // in LN mode, exactly one chunk. Wide-N projections keep 4 waves; the big-K fc2 uses more.
    int nw = (p.KT + GV_CH - 1) / GV_CH;
    if (nw < 1) nw = 1;
    if (nw > 8) nw = 8;    // 512-thread workgroups keep 256 VGPRs per lane (LN mode: d_model <= 8*6*32 = 1536)
    const int NTB = (MT == 1) ? 2 : 1;
    dim3 grid((NT_total + NTB - 1) / NTB), block(nw * 64);
    const size_t shm = sizeof(float) * ((size_t)2 * nw * MT * 16 + (size_t)nw * NTB * MT * 256);
    switch (MT) {
        case 1: gemv_dispatch_in<1, 2>(p, grid, block, shm, s); break;
        case 2: gemv_dispatch_in<2, 1>(p, grid, block, shm, s); break;
        case 3: gemv_dispatch_in<3, 1>(p, grid, block, shm, s); break;
        default: gemv_dispatch_in<4, 1>(p, grid, block, shm, s); break;
    }
}

// ------------------------------------------------------------------ causal self-attention over the KV cache
// One wave per (row, head); positions 0..pos[row]; the history of a row is read through the ancestry table.
// Flash-style over blocks of 64 positions so one rolled loop serves every length (code size is latency here: the
// two-pass form was 4.6 KiB of straight-line code, ~1.8 us of cold instruction fetch per launch):
//   per block: lane p looks up the cache row of position p (one dependent trip), then the K and the V rows are requested together — one
//   more trip —, both as lane = (position group pg, 8-dim chunk dc): 8 positions x a whole 128-byte row per load instruction (round 6; the
//   K rows were lane = position before), scores = 8 dims in the lane + a 3-step DPP butterfly over the 8 dc lanes, block max / sum by DPP,
//   and the (pg, dc) lanes accumulate their 8 dims over their 8 positions with the running-max rescale (their probabilities are in-lane).
//   tail: the 8 position groups are summed through LDS, lane d writes output dim d.
// IDENT: the rows read their history through their OWN ancestry row (every decode step; prefill rows share one): the
// ancestry row address then needs no table lookup, so the first block's cache-row lookup is requested at once, next to
// the scalar loads of the row's position, instead of behind them — one dependent trip less (of three) per launch.
// Round 3: SA_NW waves per (row, head). A wave walks every SA_NW-th block of 64 positions and the waves' (m, l, o) are merged
// through LDS in a fixed order. With one wave the blocks of a long history were a serial chain of dependent round trips
// (ancestry -> K / V -> softmax): a step at positions 225..288 (a window conditioned on the reference's full prompt) cost
// 6.4 us per layer here against 2.9 us at t = 32. Histories of <= 64 positions run exactly as before on wave 0 — the other
// waves leave at once, and ended waves do not count at the barrier.
// Round 6: EIGHT waves — one block of 64 positions per wave up to the 448-position context, no wave walks a second block. Measured twice: with
// the K rows fetched lane = position a step at positions 200 / 300 / 447 cost 398 / 412 / 425 us with 4 or 8 waves (profiles/r6b_*: the K fetch
// was what grew); with whole-row K fetches 390 / 408 / 411 us with 4 waves and 390 / 396 / 402 with 8 (profiles/r6w_step_by_position_sa8.txt) —
// now the second dependent trip is what is left. Histories of <= 256 positions are bit-identical to the 4-wave form (same blocks, same merge order).
// The decode steps (IDENT) run 8 waves; the prompt prefill (<= 228 positions, thousands of workgroups) keeps 4 — with 8 its 224-token pass
// measured 0.94 against 0.91 ms (four waves per workgroup launched only to leave).
#define SA_PF_NW 4       // the non-IDENT passes (prompt prefill, teacher-forced rows) and batched steps (> 16 rows)
#ifndef SA_NW
#define SA_NW 8          // -DSA_NW=1 / 4 (whisperlive_amd/_lib.py build_variant) = the one- / four-wave forms of the decode steps, for A/B
#endif
template <bool IDENT, int NW>
__global__ __launch_bounds__(64 * NW) void dec_self_attn2_kernel(const half_t* __restrict__ q, long ldq,
                                                                    const half_t* __restrict__ Kc,
                                                                    const half_t* __restrict__ Vc, long crs, int d,
                                                                    const int* __restrict__ pos,
                                                                    const int* __restrict__ ancrow,
                                                                    const short* __restrict__ anc,
                                                                    half_t* __restrict__ out, long ldo WLX_TR_PARAM) {
    __shared__ int crow_s[NW][64];
    __shared__ float part_s[NW][8][64];
    __shared__ float ml_s[NW][2];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = blockIdx.x, h = blockIdx.y;
    WLX_TR_BEGIN();
    // the cache rows of this wave's FIRST block, requested together with the row's position, before anything depends on either
    // (clamped to the ancestry row's 448 entries, whatever the history length turns out to be) — without it
    // the waves of the later blocks paid a second dependent round trip (position -> ancestry -> K / V)
    const short* ar = anc + (long)(IDENT ? r : ancrow[r]) * WLX_T_TEXT;
    int cr0 = 0;
    if constexpr (IDENT) cr0 = ar[(w * 64 + lane < WLX_T_TEXT) ? w * 64 + lane : WLX_T_TEXT - 1];   // (8 x 64 > 448: the last wave's lanes past the row)
    const int len = pos[r] + 1;
    const int nblk = (len + 63) >> 6;
    if (w >= nblk) return;                                  // (wave 0 always stays: len >= 1)
    int* crow = crow_s[w];
    const int pg = lane >> 3, dc = lane & 7;
    const int hoff = h * WLX_HEAD_DIM;
    float qf[8];                                            // this lane's 8 query dims (dc * 8 ..)
    {
        const f16x8 qv = ld_f16x8(q + (long)r * ldq + hoff + dc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[e] = (float)qv[e];
    }
    const int icrs = (int)crs;                             // 32-bit element offsets: cache_rows * 448 * d < 2^31
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mrun = WLX_NEG_INF, lrun = 0.f;
    WLX_TR_MARK(1);
#pragma unroll 1
    for (int p0 = w * 64; p0 < len; p0 += 64 * NW) {
        const int p = p0 + lane;
        const bool ok = p < len;
        int cr;
        if (IDENT && p0 == w * 64) cr = ok ? cr0 : __builtin_amdgcn_readlane(cr0, 0);  // (a masked lane: any valid row — the block's first position's)
        else cr = ar[ok ? p : len - 1];
        crow[lane] = cr;                                    // (visible to this same wave after the wait the reads below carry)
        // K AND V chunks, the same element offsets in both caches: lane (pg, dc) takes dims dc*8 .. dc*8+7 of positions p0 + u*8 + pg, so a
        // load instruction covers 8 positions x one whole 128-byte row each = 8 cache lines. (Through round 5 the K rows were read lane =
        // position, 16 bytes of 64 different lines per instruction, eight times over: the step grew 48 us from position 8 to 447, twice what the
        // bytes cost — profiles/r6b_step_by_position.txt.)
        f16x8 kk[8], vv[8];
        bool okp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pp = p0 + u * 8 + pg;
            okp[u] = pp < len;
            const unsigned off = (unsigned)(crow[u * 8 + pg] * icrs + (okp[u] ? pp : len - 1) * d + hoff + dc * 8);
            kk[u] = ld_f16x8(Kc + off);
            vv[u] = ld_f16x8(Vc + off);
        }
        // scores: 8 dims in the lane, then the 8 dc lanes of a position summed by a 3-step butterfly (all 8 end with the same bits)
        float sc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf((float)kk[u][e], qf[e], a);
            sc[u] = a;
        }
        WLX_DPP_SUM8x8(sc);
        float bm = WLX_NEG_INF;
#pragma unroll
        for (int u = 0; u < 8; ++u) { sc[u] = okp[u] ? sc[u] : WLX_NEG_INF; bm = fmaxf(bm, sc[u]); }
        const float mnew = fmaxf(mrun, dpp_wave_max(bm));   // finite: position p0 is valid
        const float alpha = __expf(mrun - mnew);
        float pe[8], bs = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { pe[u] = __expf(sc[u] - mnew); bs += pe[u]; }   // 0 for masked positions
        lrun = lrun * alpha + dpp_wave_sum(dc == 0 ? bs : 0.f);                      // (each position once: its dc = 0 lane)
        mrun = mnew;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pe[u], (float)vv[u][e], o[e]);
    }
    WLX_TR_MARK(2);
    // sum the 8 position groups: part[pg][dim]; lane d then owns output dim d
    *reinterpret_cast<float4*>(&part_s[w][pg][dc * 8]) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(&part_s[w][pg][dc * 8 + 4]) = make_float4(o[4], o[5], o[6], o[7]);
    float acc = 0.f;
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) acc += part_s[w][g8][lane];
    if (nblk == 1) {                                        // one block: wave 0 alone, as before
        out[(long)r * ldo + hoff + lane] = (half_t)(acc / lrun);
    } else {
        // merge the waves' partial (m, l, o) in wave order: o = sum_w o_w e^(m_w - m), l = sum_w l_w e^(m_w - m)
        part_s[w][0][lane] = acc;                           // (this wave's own row of part_s: read above by the same lanes' wave only)
        if (lane == 0) { ml_s[w][0] = mrun; ml_s[w][1] = lrun; }
        __syncthreads();                                    // the waves that left at the top do not count
        if (w == 0) {
            const int nw = nblk < NW ? nblk : NW;
            float m = ml_s[0][0];
            for (int k = 1; k < nw; ++k) m = fmaxf(m, ml_s[k][0]);
            float L = 0.f, O = 0.f;
            for (int k = 0; k < nw; ++k) {
                const float f = __expf(ml_s[k][0] - m);
                L += ml_s[k][1] * f;
                O += part_s[k][0][lane] * f;
            }
            out[(long)r * ldo + hoff + lane] = (half_t)(O / L);
        }
    }
    WLX_TR_MARK(3);
    WLX_TR_END(trc);
}

void launch_dec_self_attn(const half_t* q, long ldq, const half_t* Kc, const half_t* Vc, long crs, int d, int H,
                          const RowTables& rt, int rows, half_t* out, long ldo, const int* done, bool ident_ancestry, hipStream_t s) {
    (void)done;   // a step that runs after the search raised `done` only rewrites scratch (engine.hip decoder_pass)
    // 8 waves for the steps of one stream (<= 16 rows: up to 16 x H workgroups, the launch is one latency chain long); batched steps (hundreds of
    // rows x H workgroups) keep 4, like the prefill: their extra waves would only be launched to leave
    if (ident_ancestry && rows <= 16)
        hipLaunchKernelGGL((dec_self_attn2_kernel<true, SA_NW>), dim3(rows, H), dim3(64 * SA_NW), 0, s, q, ldq, Kc, Vc, crs, d, rt.pos,
                           rt.ancrow, rt.anc, out, ldo WLX_TR_ARG("self_attn"));
    else if (ident_ancestry)
        hipLaunchKernelGGL((dec_self_attn2_kernel<true, SA_PF_NW>), dim3(rows, H), dim3(64 * SA_PF_NW), 0, s, q, ldq, Kc, Vc, crs, d, rt.pos,
                           rt.ancrow, rt.anc, out, ldo WLX_TR_ARG("self_attn"));
    else
        hipLaunchKernelGGL((dec_self_attn2_kernel<false, SA_PF_NW>), dim3(rows, H), dim3(64 * SA_PF_NW), 0, s, q, ldq, Kc, Vc, crs, d, rt.pos,
                           rt.ancrow, rt.anc, out, ldo WLX_TR_ARG("self_attn"));
}

// ------------------------------------------------------------------ decode cross-attention (flash-decoding split over keys)
// grid (split, head, group): the R rows of an item share the item's encoder K/V, so they form ONE 16-row MFMA query
// tile (scores transposed as in attention.hip: S^T = K Q^T, softmax in-lane + 2 DPP steps, O^T = V^T P^T).
// One WORKGROUP of XA_TPS waves per (split, head, group), one 32-key tile per wave: six waves pull 8 KiB each, form
// their (m, l, O) over one tile and merge through LDS into the split's partial.
//   * K and V come TILE-PACKED from the encoder's cross-K/V GEMM epilogue (gemm.hip GEMM_CROSS_KV): per (layer, item,
//     head, 32-key tile) a 4 KiB image in MFMA operand order, so every load is one contiguous 1 KiB per wave (the
//     row-major K / transposed V of the first generations cost 12 strided loads of 32-64 byte segments per tile);
//     keys >= 1500 of the padding are zero in V (never written) and masked to -inf in the scores.
//   * the partial handed to the consumer projection is the NORMALISED fp16 O plus fp32 (m, l), the (m, l) of a row's
//     eight splits contiguous: the consumer's combine is bound by load-instruction count per CU, not by bytes.
#define XA_TPS (WLX_T_AUDIO_PAD / 32 / WLX_XSPLIT)
__global__ __launch_bounds__(XA_TPS * 64) void dec_cross_attn_kernel(const half_t* __restrict__ q, long ldq,
                                                                     const half_t* __restrict__ Kp, const half_t* __restrict__ Vp,
                                                                     long item_stride, int H, int R, int rows,
                                                                     const int* __restrict__ group_item,
                                                                     half_t* __restrict__ part_o, float* __restrict__ part_ml WLX_TR_PARAM) {
    constexpr int TPS = XA_TPS;
    static_assert(TPS * WLX_XSPLIT * 32 == WLX_T_AUDIO_PAD && TPS >= 4, "key padding must cover every split; >= 4 waves combine");
    __shared__ __attribute__((aligned(16))) float Os[TPS][16][68];
    __shared__ float MLs[TPS][16][2];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sp = blockIdx.x, h = blockIdx.y, grp = blockIdx.z;
    WLX_TR_BEGIN();
    const int item = group_item[grp];
    constexpr int T = WLX_T_AUDIO;
    const int tile = sp * TPS + wave;
    const int key0 = tile * 32;
    const long toff = (long)item * item_stride + ((long)h * (WLX_T_AUDIO_PAD / 32) + tile) * 2048 + lane * 8;
    f16x8 kf[2][2], vf[4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) kf[s2][kt] = ld_f16x8(Kp + toff + (s2 * 2 + kt) * 512);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vf[dt] = ld_f16x8(Vp + toff + dt * 512);
    int row = grp * R + c;
    const bool qok = (c < R) && (row < rows);
    if (!qok) row = grp * R;  // any valid row; result discarded
    f16x8 qf[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) qf[kt] = ld_f16x8(q + (long)row * ldq + h * WLX_HEAD_DIM + kt * 32 + g * 8);
    WLX_TR_MARK(1);

    f32x4 st[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        st[s2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) st[s2] = mfma16(kf[s2][kt], qf[kt], st[s2]);
    }
    float pv[8];
    float tmax = WLX_NEG_INF;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + s2 * 16 + g * 4 + r;
            const float v = (key < T) ? st[s2][r] : WLX_NEG_INF;
            pv[s2 * 4 + r] = v;
            tmax = fmaxf(tmax, v);
        }
    if constexpr (WLX_CQ_SWAP != 0) tmax = rows4_max(tmax);   // (v_permlane swaps instead of ds_bpermute round trips: see dec_cq_cross_attn_kernel)
    else { tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)); }
    const float msafe = (tmax == WLX_NEG_INF) ? 0.f : tmax;      // a fully masked tile: every p = exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { pv[i] = __expf(pv[i] - msafe); psum += pv[i]; }
    if constexpr (WLX_CQ_SWAP != 0) psum = rows4_sum(psum);
    else { psum += __shfl_xor(psum, 16, 64); psum += __shfl_xor(psum, 32, 64); }
    const f16x8 pf = {(half_t)pv[0], (half_t)pv[1], (half_t)pv[2], (half_t)pv[3],
                      (half_t)pv[4], (half_t)pv[5], (half_t)pv[6], (half_t)pv[7]};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const f32x4 a = mfma16(vf[dt], pf, (f32x4){0.f, 0.f, 0.f, 0.f});
        *reinterpret_cast<f32x4*>(&Os[wave][c][dt * 16 + g * 4]) = a;
    }
    if (g == 0) { MLs[wave][c][0] = tmax; MLs[wave][c][1] = psum; }
    WLX_TR_MARK(2);
    __syncthreads();
    if (wave < 4) {
        // wave dt merges output dims dt*16 .. dt*16+15 of all 16 query rows over the TPS tiles (fixed order)
        const int dt = wave;
        float mw[TPS], lw[TPS];
        float M = WLX_NEG_INF;
#pragma unroll
        for (int w = 0; w < TPS; ++w) { mw[w] = MLs[w][c][0]; lw[w] = MLs[w][c][1]; M = fmaxf(M, mw[w]); }
        float l = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < TPS; ++w) {
            const float e = (mw[w] == WLX_NEG_INF) ? 0.f : __expf(mw[w] - M);
            l += e * lw[w];
            const f32x4 t = *reinterpret_cast<const f32x4*>(&Os[w][c][dt * 16 + g * 4]);
            o[0] += e * t[0]; o[1] += e * t[1]; o[2] += e * t[2]; o[3] += e * t[3];
        }
        const long ih = (long)grp * H + h;
        const float inv = 1.0f / l;               // every split starts below key 1500: l > 0
        const f16x4 hv = {(half_t)(o[0] * inv), (half_t)(o[1] * inv), (half_t)(o[2] * inv), (half_t)(o[3] * inv)};
        *reinterpret_cast<f16x4*>(part_o + ((ih * WLX_XSPLIT + sp) * 16 + c) * 64 + dt * 16 + g * 4) = hv;
        if (dt == 0 && g == 0) *reinterpret_cast<float2*>(part_ml + (ih * 16 + c) * (WLX_XSPLIT * 2) + sp * 2) = make_float2(M, l);
    }
    WLX_TR_MARK(3);
    WLX_TR_END(trc);
}

// ------------------------------------------------------------------ fused LayerNorm + cross-attention query + cross-attention
// One launch instead of two (LN2 + q projection, then attention): the workgroup of (split, head, group) computes the
// 64 query columns of ITS head for the group's <= 16 rows itself — LayerNorm of the rows (one wave per row, DPP), the
// head's 4 n-tiles x KT k-tiles of Wcq streamed by its six waves (KT/6 k-tiles x 4 n-tiles each), K-reduction through
// LDS — and goes straight on to its 32-key tiles. The eight split-workgroups of a head repeat the head's 96 KiB of Wcq;
// the blockIdx -> (head, split) map puts them on ONE XCD (workgroup ids are dealt round-robin over the 8 XCDs), so the
// repeats are L2 hits, not HBM reads. What it buys: one launch boundary (1.6 us) and one launch's fixed latency per
// layer; what it costs: ~200 load instructions per CU instead of ~60 (2.2 us of issue at ~11 ns each).
// Eligibility (launcher): d_model = 256 * LNV with KT % 6 == 0 — Whisper-small; other sizes keep the two launches.
template <int LNV, int KPW>
__global__ __launch_bounds__(XA_TPS * 64) void dec_cq_cross_attn_kernel(
    const float* __restrict__ X, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
    const half_t* __restrict__ Wp, const float* __restrict__ bias, float qscale, int KT,
    const half_t* __restrict__ Kp, const half_t* __restrict__ Vp, long item_stride, int H, int R, int rows,
    const int* __restrict__ group_item, half_t* __restrict__ part_o, float* __restrict__ part_ml WLX_TR_PARAM) {
    constexpr int TPS = XA_TPS;
    static_assert(TPS == 6, "six waves: KT/6 k-tiles of the query projection and one key tile each");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WLX_TR_BEGIN();
    // ---- (head, split) of this workgroup: the 8 splits of a head share blockIdx % 8, i.e. one XCD
    const int per_grp = H * WLX_XSPLIT;
    const int grp = blockIdx.x / per_grp, wg = blockIdx.x - grp * per_grp;
    int h, sp;
    {
        const int x = wg & 7, j = wg >> 3, full = H >> 3, rem = H & 7;
        if (rem == 0 || rem == 4) {
            if (j < 8 * full) { h = (j >> 3) * 8 + x; sp = j & 7; }
            else { h = 8 * full + (x >> 1); sp = (x & 1) * 4 + (j - 8 * full); }
        } else { h = wg >> 3; sp = wg & 7; }
    }
    const int d = KT * 32;
    const int ldxs = d + 8;
    // LDS: [0, xs_bytes) fp16 LayerNorm rows; then a region shared in time by the K-reduction partials and the
    // attention merge buffers; then the 16 x 64 fp16 query tile
    half_t* xs = reinterpret_cast<half_t*>(smem);
    float* regB = smem + ((R * ldxs * 2 + 15) / 16) * 4;
    float* accred = regB;                                                  // [6][4][64][4]
    float (*Os)[16][68] = reinterpret_cast<float (*)[16][68]>(regB);       // [6][16][68]
    float* MLs = regB + TPS * 16 * 68;                                     // [6][16][2]
    half_t* qs = reinterpret_cast<half_t*>(MLs + TPS * 16 * 2);            // [16][72]

    // ---- everything this wave will ever load, requested up front — the LayerNorm row of the first trip FIRST (round 2):
    // vmcnt retires in order, so with the 24 KiB of weights / K / V per wave ahead of it the row arrived last and the
    // LayerNorm (which everything else waits for) started 2.6 us into a 5.9 us launch
    const int nrow = (rows - grp * R < R) ? rows - grp * R : R;           // live rows of this group
    float4 x0[LNV];
    {
        const int r0 = (wave < nrow) ? wave : nrow - 1;
        const float4* x4 = reinterpret_cast<const float4*>(X + (long)(grp * R + r0) * ldx) + lane;
#pragma unroll
        for (int j = 0; j < LNV; ++j) x0[j] = x4[64 * j];
    }
    // gamma / beta (round 6, WLX_CQ_GB_LDS; -DWLX_CQ_GB_LDS=0 = every wave loads all of both, for A/B): ONE KiB piece of [gamma | beta] per wave
    // (six waves, 2 x 3 pieces), shared through LDS before the first row is normalised — 30 of the workgroup's 204 wave-level loads less in front
    // of its weights and K / V, and this launch is bound by exactly that count (a CU retires one per ~11 ns, L1 hit or not). Same values into the
    // same arithmetic: bit-identical. In-kernel timeline (profiles/r6ad_*): rows normalised 0.26 us LATER (the exchange's barrier), query tile
    // ready 0.35 us earlier, workgroup 5.06 -> 4.76 us. The same exchange in dec_gemv2_kernel's LayerNorm prologues measured the other way — first
    // projection 2.04 -> 2.27 us, first MLP projection 1.90 -> 2.31 us per workgroup, step graph +11 us: there the weights are 24 of 72-120 loads
    // and a wave's normalisation waited for nobody else's loads (scripts/patches/r6ad_*.diff) — so it is used here only.
    float4 gq[LNV], bq[LNV];
    float4 gbp = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (WLX_CQ_GB_LDS != 0) {
        static_assert(WLX_CQ_GB_LDS == 0 || 2 * LNV == TPS, "one [gamma | beta] piece per wave");
        const float* src = (wave < LNV) ? gamma + wave * 256 : beta + (wave - LNV) * 256;
        gbp = reinterpret_cast<const float4*>(src)[lane];
    } else {
        const float4* g4 = reinterpret_cast<const float4*>(gamma) + lane;
        const float4* b4 = reinterpret_cast<const float4*>(beta) + lane;
#pragma unroll
        for (int j = 0; j < LNV; ++j) { gq[j] = g4[64 * j]; bq[j] = b4[64 * j]; }
    }
    asm volatile("" ::: "memory");     // compile-time fence: keep the requests below behind the ones above
    const int kw0 = wave * KPW;
    const half_t* wp = Wp + ((long)(h * 4) * KT + kw0) * 512 + lane * 8;
    f16x8 wf[KPW][4];
#pragma unroll
    for (int j = 0; j < KPW; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[j][i] = ld_nt_f16x8(wp + (long)i * KT * 512 + j * 512);
    // (Tried: consuming this scalar load through an asm barrier placed here, so that the wave does not wait for it between the LayerNorm operands'
    // requests and the weights' — hipcc then sinks the ROW loads behind the weights: the order the round-2 reordering removed. Left alone.)
    const int item = group_item[grp];
    const int tile = sp * TPS + wave;
    const int key0 = tile * 32;
    const long toff = (long)item * item_stride + ((long)h * (WLX_T_AUDIO_PAD / 32) + tile) * 2048 + lane * 8;
    f16x8 kf[2][2], vf[4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) kf[s2][kt] = ld_f16x8(Kp + toff + (s2 * 2 + kt) * 512);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vf[dt] = ld_f16x8(Vp + toff + dt * 512);
    const float4 bq4 = *reinterpret_cast<const float4*>(bias + h * 64 + (wave & 3) * 16 + g * 4);
    {   // LayerNorm: wave w normalises rows w, w + 6, ... of the group
        constexpr float invK = 1.0f / (256.0f * LNV);
        auto ln_row = [&](float4 (&x)[LNV], int r, bool keep) {
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < LNV; ++j) sm += (x[j].x + x[j].y) + (x[j].z + x[j].w);
            const float mean = dpp_wave_sum(sm) * invK;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < LNV; ++j) {
                x[j].x -= mean; x[j].y -= mean; x[j].z -= mean; x[j].w -= mean;
                q += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
            }
            const float rstd = rsqrtf(dpp_wave_sum(q) * invK + 1e-5f);
            half_t* dst = xs + r * ldxs + lane * 4;
#pragma unroll
            for (int j = 0; j < LNV; ++j) {
                const f16x4 hv = {(half_t)(x[j].x * rstd * gq[j].x + bq[j].x), (half_t)(x[j].y * rstd * gq[j].y + bq[j].y),
                                  (half_t)(x[j].z * rstd * gq[j].z + bq[j].z), (half_t)(x[j].w * rstd * gq[j].w + bq[j].w)};
                if (keep) *reinterpret_cast<f16x4*>(dst + 256 * j) = hv;
            }
        };
        if constexpr (WLX_CQ_GB_LDS != 0) {
            float4* gbs4 = reinterpret_cast<float4*>(qs + 16 * 72);       // [2 LNV][64] float4 behind the query tile
            gbs4[wave * 64 + lane] = gbp;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < LNV; ++j) { gq[j] = gbs4[j * 64 + lane]; bq[j] = gbs4[(LNV + j) * 64 + lane]; }
        }
        // first trip: straight-line and UNCONDITIONAL (a wave without a row normalises the clamped row it loaded and
        // keeps nothing): inside an `if (wave < nrow)` hipcc sinks the row loads into the branch, behind the weights
        ln_row(x0, (wave < nrow) ? wave : nrow - 1, wave < nrow);
#pragma unroll 1
        for (int r = wave + TPS; r < nrow; r += TPS) {                      // groups of more than six rows (prefill)
            const float4* x4 = reinterpret_cast<const float4*>(X + (long)(grp * R + r) * ldx) + lane;
            float4 x[LNV];
#pragma unroll
            for (int j = 0; j < LNV; ++j) x[j] = x4[64 * j];
            ln_row(x, r, true);
        }
    }
    WLX_TR_MARK(1);
    __syncthreads();
    // ---- the head's query columns: this wave's K slice of all 4 n-tiles
    const int crow = (c < nrow) ? c : nrow - 1;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const half_t* xr = xs + crow * ldxs + kw0 * 32 + g * 8;
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            const f16x8 xf = *reinterpret_cast<const f16x8*>(xr + j * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma16(wf[j][i], xf, acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(accred + ((wave * 4 + i) * 64 + lane) * 4) = acc[i];
    __syncthreads();
    if (wave < 4) {      // wave i finishes n-tile i: fixed-order sum over the six K slices, bias, q scale -> fp16 query tile
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < TPS; ++w) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(accred + ((w * 4 + wave) * 64 + lane) * 4);
            v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
        const f16x4 hv = {(half_t)((v[0] + bq4.x) * qscale), (half_t)((v[1] + bq4.y) * qscale),
                          (half_t)((v[2] + bq4.z) * qscale), (half_t)((v[3] + bq4.w) * qscale)};
        *reinterpret_cast<f16x4*>(qs + c * 72 + wave * 16 + g * 4) = hv;     // query row c, head dims wave*16 + g*4 ..
    }
    WLX_TR_MARK(2);
    __syncthreads();                                                        // qs complete; accred dead (Os aliases it)
    f16x8 qf[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) qf[kt] = *reinterpret_cast<const f16x8*>(qs + c * 72 + kt * 32 + g * 8);

    // ---- attention over this wave's tile: identical to dec_cross_attn_kernel from here on
    constexpr int T = WLX_T_AUDIO;
    f32x4 st[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        st[s2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) st[s2] = mfma16(kf[s2][kt], qf[kt], st[s2]);
    }
    float pv[8];
    float tmax = WLX_NEG_INF;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + s2 * 16 + g * 4 + r;
            const float v = (key < T) ? st[s2][r] : WLX_NEG_INF;
            pv[s2 * 4 + r] = v;
            tmax = fmaxf(tmax, v);
        }
    // (max / sum over the four 16-lane rows by v_permlane16/32_swap, round 6: the ds_bpermute form cost four dependent LDS round trips in the tail
    // of the launch; same values — max is exact, the sum's additions commute — WLX_CQ_SWAP=0 = the ds_bpermute form, for A/B)
    if constexpr (WLX_CQ_SWAP != 0) tmax = rows4_max(tmax);
    else { tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)); }
    const float msafe = (tmax == WLX_NEG_INF) ? 0.f : tmax;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { pv[i] = __expf(pv[i] - msafe); psum += pv[i]; }
    if constexpr (WLX_CQ_SWAP != 0) psum = rows4_sum(psum);
    else { psum += __shfl_xor(psum, 16, 64); psum += __shfl_xor(psum, 32, 64); }
    const f16x8 pf = {(half_t)pv[0], (half_t)pv[1], (half_t)pv[2], (half_t)pv[3],
                      (half_t)pv[4], (half_t)pv[5], (half_t)pv[6], (half_t)pv[7]};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const f32x4 a = mfma16(vf[dt], pf, (f32x4){0.f, 0.f, 0.f, 0.f});
        *reinterpret_cast<f32x4*>(&Os[wave][c][dt * 16 + g * 4]) = a;
    }
    if (g == 0) { MLs[(wave * 16 + c) * 2] = tmax; MLs[(wave * 16 + c) * 2 + 1] = psum; }
    __syncthreads();
    if (wave < 4) {
        const int dt = wave;
        float mw[TPS], lw[TPS];
        float M = WLX_NEG_INF;
#pragma unroll
        for (int w = 0; w < TPS; ++w) { mw[w] = MLs[(w * 16 + c) * 2]; lw[w] = MLs[(w * 16 + c) * 2 + 1]; M = fmaxf(M, mw[w]); }
        float l = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < TPS; ++w) {
            const float e = (mw[w] == WLX_NEG_INF) ? 0.f : __expf(mw[w] - M);
            l += e * lw[w];
            const f32x4 t = *reinterpret_cast<const f32x4*>(&Os[w][c][dt * 16 + g * 4]);
            o[0] += e * t[0]; o[1] += e * t[1]; o[2] += e * t[2]; o[3] += e * t[3];
        }
        const long ih = (long)grp * H + h;
        const float inv = 1.0f / l;
        const f16x4 hv = {(half_t)(o[0] * inv), (half_t)(o[1] * inv), (half_t)(o[2] * inv), (half_t)(o[3] * inv)};
        *reinterpret_cast<f16x4*>(part_o + ((ih * WLX_XSPLIT + sp) * 16 + c) * 64 + dt * 16 + g * 4) = hv;
        if (dt == 0 && g == 0) *reinterpret_cast<float2*>(part_ml + (ih * 16 + c) * (WLX_XSPLIT * 2) + sp * 2) = make_float2(M, l);
    }
    WLX_TR_MARK(3);
    WLX_TR_END(trc);
}

// eligibility: d_model 768 (LNV = 3, KT = 24 = 6 waves x 4 k-tiles), groups of <= 16 rows; WLX_NO_FUSED_CQ=1 forces the
// two separate launches (A/B)
bool dec_cq_cross_attn_eligible(int d, int H, int R) {
    static const bool off = [] { const char* e = wlx_ab("WLX_NO_FUSED_CQ"); return e && e[0] == '1'; }();
    if (off || g_decode_v1 || d != 768 || H * 64 != d || R < 1 || R > 16) return false;
    const size_t xs_floats = (size_t)((R * (d + 8) * 2 + 15) / 16) * 4;
    const size_t shm = sizeof(float) * (xs_floats + (size_t)XA_TPS * 16 * 68 + XA_TPS * 16 * 2) + 16 * 72 * sizeof(half_t) + (WLX_CQ_GB_LDS != 0 ? 6 * 1024 : 0);
    return shm <= 64 * 1024;
}
void launch_dec_cq_cross_attn(const float* X, long ldx, const float* gamma, const float* beta, const half_t* Wp, const float* bias,
                              float qscale, int d, const half_t* Kp, const half_t* Vp, long item_stride, int H, int R, int groups,
                              int rows, const int* group_item, half_t* part_o, float* part_ml, hipStream_t s) {
    const int KT = d / 32;
    const size_t xs_floats = (size_t)((R * (d + 8) * 2 + 15) / 16) * 4;
    const size_t shm = sizeof(float) * (xs_floats + (size_t)XA_TPS * 16 * 68 + XA_TPS * 16 * 2) + 16 * 72 * sizeof(half_t) + (WLX_CQ_GB_LDS != 0 ? 6 * 1024 : 0);
    hipLaunchKernelGGL((dec_cq_cross_attn_kernel<3, 4>), dim3(H * WLX_XSPLIT * groups), dim3(XA_TPS * 64), shm, s, X, ldx, gamma, beta,
                       Wp, bias, qscale, KT, Kp, Vp, item_stride, H, R, rows, group_item, part_o, part_ml WLX_TR_ARG("cq_cross_attn"));
}

// ------------------------------------------------------------------ split combine of the cross attention, on its own
// For batched rows (> 16) the fused form — every workgroup of the output projection combining ALL rows' eight split
// partials in its prologue — costs 12 us of a 14.6 us launch at 40 rows x 20 heads (80 workgroups each looping 13 times
// over 6400 (row, head, 8-dim) items; profiles/r2s decode-step timeline). Here one thread per item does it once, the
// result goes to the fp16 attention rows and the projection runs as a plain fp16-rows-in launch. (Up to 16 rows the fused
// form stays: one trip, no extra launch.)
__global__ __launch_bounds__(256) void dec_xattn_combine_kernel(const half_t* __restrict__ part_o, const float* __restrict__ part_ml,
                                                                int M, int H, int R, half_t* __restrict__ out, long ldo WLX_TR_PARAM) {
    WLX_TR_BEGIN();
    const int it0 = blockIdx.x * 256 + threadIdx.x;
    const bool live = it0 < M * H * 8;
    const int it = live ? it0 : M * H * 8 - 1;              // (clamped: every thread runs the same code; only live ones store)
    const int q8 = it & 7, hm = it >> 3;
    const int m = hm / H, hh = hm - m * H;
    const int item = m / R, qi = m - item * R;
    const long ih = (long)item * H + hh;
    const float4* mlp = reinterpret_cast<const float4*>(part_ml + (ih * 16 + qi) * (WLX_XSPLIT * 2));
    const half_t* op = part_o + (ih * WLX_XSPLIT * 16 + qi) * 64 + q8 * 8;
    float4 ml[WLX_XSPLIT / 2];
    f16x8 ov[WLX_XSPLIT];
#pragma unroll
    for (int sp = 0; sp < WLX_XSPLIT / 2; ++sp) ml[sp] = mlp[sp];
#pragma unroll
    for (int sp = 0; sp < WLX_XSPLIT; ++sp) ov[sp] = ld_f16x8(op + sp * 1024);
    float mmax = fmaxf(ml[0].x, ml[0].z);
#pragma unroll
    for (int sp = 1; sp < WLX_XSPLIT / 2; ++sp) mmax = fmaxf(mmax, fmaxf(ml[sp].x, ml[sp].z));
    float den = 0.f;
    float num[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < WLX_XSPLIT; ++sp) {        // (the arithmetic and its order are the fused combine's: identical rows)
        const float mm = (sp & 1) ? ml[sp >> 1].z : ml[sp >> 1].x, ll = (sp & 1) ? ml[sp >> 1].w : ml[sp >> 1].y;
        const float w = __expf(mm - mmax) * ll;
        den += w;
#pragma unroll
        for (int e = 0; e < 8; ++e) num[e] += w * (float)ov[sp][e];
    }
    const float inv = 1.0f / den;
    const f16x8 hv = {(half_t)(num[0] * inv), (half_t)(num[1] * inv), (half_t)(num[2] * inv), (half_t)(num[3] * inv),
                      (half_t)(num[4] * inv), (half_t)(num[5] * inv), (half_t)(num[6] * inv), (half_t)(num[7] * inv)};
    if (live) *reinterpret_cast<f16x8*>(out + (long)m * ldo + hh * 64 + q8 * 8) = hv;
    WLX_TR_END(trc);
}
void launch_dec_xattn_combine(const half_t* part_o, const float* part_ml, int M, int H, int R, half_t* out, long ldo, hipStream_t s) {
    hipLaunchKernelGGL(dec_xattn_combine_kernel, dim3((M * H * 8 + 255) / 256), dim3(256), 0, s, part_o, part_ml, M, H, R, out, ldo WLX_TR_ARG("xattn_combine"));
}

// ------------------------------------------------------------------ cross-attention scores of one head, for word alignment
// (ctranslate2 Whisper.align, called from transcriber_faster_whisper.py:1657: the QK of the alignment heads over a
// teacher-forced pass). Raw scores q.k (q already carries head_dim^-0.5) of `rows` query rows against the 1536 padded
// keys of one (item, head), written as fp32 [rows][1536]; the softmax over the first num_frames/2 keys, the
// normalisation, median filter and DTW run on the host side of wlx_align (they are O(tokens x 1500) scalar work).
// grid (WLX_XSPLIT key splits, row tiles of 16); XA_TPS waves per workgroup, one tile-packed 32-key tile each.
__global__ __launch_bounds__(XA_TPS * 64) void dec_align_scores_kernel(const half_t* __restrict__ q, long ldq,
                                                                       const half_t* __restrict__ Kp, int h, int rows,
                                                                       float* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = tid >> 6;
    const int tile = blockIdx.x * XA_TPS + wave;
    const int row0 = blockIdx.y * 16;
    const long toff = ((long)h * (WLX_T_AUDIO_PAD / 32) + tile) * 2048 + lane * 8;
    f16x8 kf[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) kf[s2][kt] = ld_f16x8(Kp + toff + (s2 * 2 + kt) * 512);
    int row = row0 + c;
    const bool ok = row < rows;
    if (!ok) row = rows - 1;
    f16x8 qf[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) qf[kt] = ld_f16x8(q + (long)row * ldq + h * WLX_HEAD_DIM + kt * 32 + g * 8);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) st = mfma16(kf[s2][kt], qf[kt], st);
        // st[r] = score of key tile*32 + s2*16 + g*4 + r for query row c
        if (ok) *reinterpret_cast<f32x4*>(out + (long)row * WLX_T_AUDIO_PAD + tile * 32 + s2 * 16 + g * 4) = st;
    }
}

void launch_dec_align_scores(const half_t* q, long ldq, const half_t* Kp_item, int h, int rows, float* out, hipStream_t s) {
    hipLaunchKernelGGL(dec_align_scores_kernel, dim3(WLX_XSPLIT, (rows + 15) / 16), dim3(XA_TPS * 64), 0, s, q, ldq, Kp_item, h, rows, out);
}

void launch_dec_cross_attn(const half_t* q, long ldq, const half_t* Kp, const half_t* Vp, long item_stride, int H, int R,
                           int groups, int rows, const int* group_item, half_t* part_o, float* part_ml, hipStream_t s) {
    hipLaunchKernelGGL(dec_cross_attn_kernel, dim3(WLX_XSPLIT, H, groups), dim3(XA_TPS * 64), 0, s, q, ldq, Kp, Vp, item_stride,
                       H, R, rows, group_item, part_o, part_ml WLX_TR_ARG("cross_attn"));
}

}  // namespace wlx
