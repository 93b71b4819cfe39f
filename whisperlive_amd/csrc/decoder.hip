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
            // combine the WLX_XSPLIT partial (m, l, O) triples of the cross attention
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
                        const long pb = ((long)item * p.H + h) * WLX_XSPLIT;
                        float ms[WLX_XSPLIT], ls[WLX_XSPLIT];
                        float mmax = WLX_NEG_INF;
#pragma unroll
                        for (int sp = 0; sp < WLX_XSPLIT; ++sp) {
                            float2 ml = *reinterpret_cast<const float2*>(p.part_ml + ((pb + sp) * 16 + qi) * 2);
                            ms[sp] = ml.x; ls[sp] = ml.y;
                            mmax = fmaxf(mmax, ml.x);
                        }
                        float den = 0.f;
                        float num[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int sp = 0; sp < WLX_XSPLIT; ++sp) {
                            const float w = __expf(ms[sp] - mmax);
                            den += w * ls[sp];
                            const float4* op = reinterpret_cast<const float4*>(p.part_o + ((pb + sp) * 16 + qi) * 64 + dd);
                            float4 a = op[0], b = op[1];
                            num[0] += w * a.x; num[1] += w * a.y; num[2] += w * a.z; num[3] += w * a.w;
                            num[4] += w * b.x; num[5] += w * b.y; num[6] += w * b.z; num[7] += w * b.w;
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


// ------------------------------------------------------------------ second-generation skinny GEMM, M <= 16 rows
// The decode step of ONE stream (beam rows only) is a chain of ~100 dependent launches of 1-5 MB each, so a
// launch is bound by the LENGTH OF ITS DEPENDENT-LOAD CHAIN, not by bandwidth. This kernel therefore
//   * issues every global load it will ever need (done flag, weight fragments [non-temporal: streamed
//     once], activations, LayerNorm gamma/beta, bias, residual, KV-cache row tables) up front, in one
//     burst, before the first wait — one HBM round trip per launch instead of five;
//   * keeps <= 6 KiB of weights per wave and one 16-column n-tile per workgroup (two for the vocabulary
//     projection), so a projection spreads over 48..192 CUs (3242/2 workgroups for the logits) instead of
//     24, and the big-K fc2 runs 16 waves per workgroup instead of 24 KiB per wave;
//   * combines the split cross-attention partials cooperatively per wave (each wave owns the heads of
//     its K slice) through LDS instead of 144 dependent loads per lane.
// Arithmetic (MFMA operand order, fixed-order LDS K reduction, epilogue) is identical to dec_gemv_kernel.
__device__ __forceinline__ f16x8 ld_nt_f16x8(const half_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p));
}

template <int NTB, int IN, int OUT, int MAXT>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(MAXT / 256, MAXT / 256))) void dec_gemv1_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int nw = blockDim.x >> 6;
    const int KT = p.KT, KTW = p.KTW;
    const int kt0 = wave * KTW;
    const int kt1 = (kt0 + KTW < KT) ? kt0 + KTW : KT;
    const int NT_total = (p.N + 15) >> 4;
    WLX_TR_BEGIN();

    int dn = 0;
    if (p.done) dn = *p.done;

    // ---- weights first
    const half_t* wbase[NTB];
#pragma unroll
    for (int i = 0; i < NTB; ++i) {
        int nt = blockIdx.x * NTB + i;
        if (nt >= NT_total) nt = NT_total - 1;
        wbase[i] = p.Wp + ((long)nt * KT * 64 + lane) * 8;
    }
    f16x8 wf[GV_CH][NTB];
#pragma unroll
    for (int j = 0; j < GV_CH; ++j) {
        int kt = kt0 + j;
        if (kt > KT - 1) kt = KT - 1;
#pragma unroll
        for (int i = 0; i < NTB; ++i) wf[j][i] = ld_nt_f16x8(wbase[i] + (long)kt * 512);
    }

    // ---- epilogue operands of the waves that will finish a tile (wave i < NTB owns tile i). Loaded by every lane
    // from clamped (always valid) addresses: an unconditional load carries no wait until its first use.
    const int ntile_e = blockIdx.x * NTB + wave;
    const bool rowok = c < p.M;
    const int crow = rowok ? c : 0;
    const bool epi = (wave < NTB) && (ntile_e < NT_total) && rowok;
    const int n_e = ntile_e * 16 + g * 4;
    const int n_ld = ((ntile_e < NT_total) ? ntile_e : NT_total - 1) * 16 + g * 4;
    float4 bias_e = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 res_e = make_float4(0.f, 0.f, 0.f, 0.f);
    int rc_e = 0, rp_e = 0;
    if constexpr (OUT != GEMV_OUT_F32) bias_e = *reinterpret_cast<const float4*>(p.bias + n_ld);   // biased layers: N % 16 == 0
    if constexpr (OUT == GEMV_OUT_RESID) res_e = *reinterpret_cast<const float4*>(p.Xres + (long)crow * p.ldxres + n_ld);
    if constexpr (OUT == GEMV_OUT_QKV) { rc_e = p.row_cache[crow]; rp_e = p.row_pos[crow]; }

    float* red = smem;                       // [2][nw][16]
    float* accred = smem + 2 * nw * 16;      // [nw][NTB][64][4]
    f16x8 xf[GV_CH];

    if constexpr (IN == GEMV_IN_LN) {
        // Cooperative LayerNorm over K = d_model. Every workgroup needs LN(x) of all M rows as MFMA B fragments; loading
        // x / gamma / beta straight into fragment layout costs 36 x 16 B per lane (each of the 16 row-lanes re-reads
        // gamma and beta, and rows are replicated across g), ~150 KB of L1 traffic per workgroup for 24 KB of weights.
        // Instead wave w normalises rows w, w + nw, ... with lane-contiguous float4 loads (a row = K/4 float4, lane l
        // holds l, l + 64, ...), reduces mean / variance with wave shuffles only (no barrier), and writes the fp16 row to
        // LDS; after ONE barrier each wave reads its B fragments back (row stride K + 8 halfs: conflict-free b128 reads).
        constexpr int LNV = 6;                                  // float4 per lane per row: K <= 1536
        const int ldxs = p.K + 8;
        half_t* xs = reinterpret_cast<half_t*>(accred + nw * NTB * 256);
        const int nv = p.K >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(p.gamma);
        const float4* b4 = reinterpret_cast<const float4*>(p.beta);
        float4 gq[LNV], bq[LNV];
#pragma unroll
        for (int j = 0; j < LNV; ++j) {
            const int idx = lane + 64 * j;
            const int idc = (idx < nv) ? idx : 0;
            gq[j] = g4[idc]; bq[j] = b4[idc];
        }
        if (dn) return;
        const float invK = 1.0f / (float)p.K;
        for (int r0 = wave; r0 < p.M; r0 += 2 * nw) {           // two rows per trip: M <= 2 nw needs one trip
            const int r1 = r0 + nw;
            const bool has1 = r1 < p.M;
            const float4* xa4 = reinterpret_cast<const float4*>(p.X + (long)r0 * p.ldx);
            const float4* xb4 = reinterpret_cast<const float4*>(p.X + (long)(has1 ? r1 : r0) * p.ldx);
            float4 xa[LNV], xb[LNV];
#pragma unroll
            for (int j = 0; j < LNV; ++j) {
                const int idx = lane + 64 * j;
                const int idc = (idx < nv) ? idx : 0;
                xa[j] = xa4[idc]; xb[j] = xb4[idc];
            }
            if (r0 == wave) WLX_TR_MARK(1);
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int j = 0; j < LNV; ++j)
                if (lane + 64 * j < nv) {
                    sa += (xa[j].x + xa[j].y) + (xa[j].z + xa[j].w);
                    sb += (xb[j].x + xb[j].y) + (xb[j].z + xb[j].w);
                }
            const float ma = wave_sum(sa) * invK, mb = wave_sum(sb) * invK;
            float qa = 0.f, qb = 0.f;
#pragma unroll
            for (int j = 0; j < LNV; ++j)
                if (lane + 64 * j < nv) {
                    float t;
                    t = xa[j].x - ma; qa += t * t; t = xa[j].y - ma; qa += t * t; t = xa[j].z - ma; qa += t * t; t = xa[j].w - ma; qa += t * t;
                    t = xb[j].x - mb; qb += t * t; t = xb[j].y - mb; qb += t * t; t = xb[j].z - mb; qb += t * t; t = xb[j].w - mb; qb += t * t;
                }
            const float ra = rsqrtf(wave_sum(qa) * invK + 1e-5f), rb = rsqrtf(wave_sum(qb) * invK + 1e-5f);
#pragma unroll
            for (int j = 0; j < LNV; ++j) {
                const int idx = lane + 64 * j;
                if (idx < nv) {
                    const f16x4 ha = {(half_t)((xa[j].x - ma) * ra * gq[j].x + bq[j].x), (half_t)((xa[j].y - ma) * ra * gq[j].y + bq[j].y),
                                      (half_t)((xa[j].z - ma) * ra * gq[j].z + bq[j].z), (half_t)((xa[j].w - ma) * ra * gq[j].w + bq[j].w)};
                    *reinterpret_cast<f16x4*>(xs + (long)r0 * ldxs + idx * 4) = ha;
                    if (has1) {
                        const f16x4 hb = {(half_t)((xb[j].x - mb) * rb * gq[j].x + bq[j].x), (half_t)((xb[j].y - mb) * rb * gq[j].y + bq[j].y),
                                          (half_t)((xb[j].z - mb) * rb * gq[j].z + bq[j].z), (half_t)((xb[j].w - mb) * rb * gq[j].w + bq[j].w)};
                        *reinterpret_cast<f16x4*>(xs + (long)r1 * ldxs + idx * 4) = hb;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < GV_CH; ++j) {
            f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kt0 + j < kt1 && rowok) o = *reinterpret_cast<const f16x8*>(xs + (long)c * ldxs + (kt0 + j) * 32 + g * 8);
            xf[j] = o;
        }
    } else if constexpr (IN == GEMV_IN_XATTN) {
        // wave-local combine of the WLX_XSPLIT (m, l, O) partials of the heads inside this wave's K slice
        // (KTW even => whole heads), written as fp16 rows to this wave's LDS image, then read back as B fragments.
        if (dn) return;
        WLX_TR_MARK(1);
        const int ldxs = KTW * 32;
        half_t* xs = reinterpret_cast<half_t*>(smem + 2 * nw * 16 + nw * NTB * 256) + (long)wave * 16 * ldxs;
        const int nh = (kt1 - kt0) >> 1;
        const int h0 = kt0 >> 1;
        const int n_it = p.M * nh * 16;     // (row m, head hh, 4-float group q4)
        // two items per lane per trip: 32 independent loads in flight (one L2 round trip per 128 items)
        for (int it0 = lane; it0 < n_it; it0 += 128) {
            float2 ml[2][WLX_XSPLIT];
            float4 ov[2][WLX_XSPLIT];
            int mrow[2], hcol[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = it0 + u * 64;
                const int itc = (it < n_it) ? it : it0;
                const int q4 = itc & 15;
                const int t2 = itc >> 4;
                const int m = t2 / nh, hh = t2 - m * nh;
                const int item = m / p.R, qi = m - item * p.R;
                const long pb = ((long)item * p.H + h0 + hh) * WLX_XSPLIT;
                mrow[u] = m; hcol[u] = hh * 64 + q4 * 4;
#pragma unroll
                for (int sp = 0; sp < WLX_XSPLIT; ++sp) {
                    ml[u][sp] = *reinterpret_cast<const float2*>(p.part_ml + ((pb + sp) * 16 + qi) * 2);
                    ov[u][sp] = *reinterpret_cast<const float4*>(p.part_o + ((pb + sp) * 16 + qi) * 64 + q4 * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float mmax = WLX_NEG_INF;
#pragma unroll
                for (int sp = 0; sp < WLX_XSPLIT; ++sp) mmax = fmaxf(mmax, ml[u][sp].x);
                float den = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
#pragma unroll
                for (int sp = 0; sp < WLX_XSPLIT; ++sp) {
                    const float w = __expf(ml[u][sp].x - mmax);
                    den += w * ml[u][sp].y;
                    n0 += w * ov[u][sp].x; n1 += w * ov[u][sp].y; n2 += w * ov[u][sp].z; n3 += w * ov[u][sp].w;
                }
                const float inv = 1.0f / den;
                const f16x4 hv = {(half_t)(n0 * inv), (half_t)(n1 * inv), (half_t)(n2 * inv), (half_t)(n3 * inv)};
                // an out-of-range slot recomputes item it0 and rewrites the same value: no branch, so the 32 loads stay hoisted
                *reinterpret_cast<f16x4*>(xs + (long)mrow[u] * ldxs + hcol[u]) = hv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < GV_CH; ++j) {
            f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kt0 + j < kt1 && rowok) o = *reinterpret_cast<const f16x8*>(xs + (long)c * ldxs + j * 32 + g * 8);
            xf[j] = o;
        }
    }

    f32x4 acc[NTB];
#pragma unroll
    for (int i = 0; i < NTB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (IN != GEMV_IN_F16) WLX_TR_MARK(2);

    if constexpr (IN == GEMV_IN_F16) {
        // activations are fp16 rows already; the loop streams further weight chunks when KTW > GV_CH
        f16x8 xn[GV_CH];
#pragma unroll
        for (int j = 0; j < GV_CH; ++j) {
            int kt = kt0 + j;
            if (kt > KT - 1) kt = KT - 1;
            xn[j] = ld_f16x8(p.Xh + (long)crow * p.ldxh + kt * 32 + g * 8);
        }
        if (dn) return;
        WLX_TR_MARK(1);
        WLX_TR_MARK_NOWAIT(2);
        for (int base = kt0; base < kt1; base += GV_CH) {
            f16x8 wn[GV_CH][NTB], xn2[GV_CH];
            const bool more = base + GV_CH < kt1;
            if (more) {
#pragma unroll
                for (int j = 0; j < GV_CH; ++j) {
                    int kt = base + GV_CH + j;
                    if (kt > KT - 1) kt = KT - 1;
#pragma unroll
                    for (int i = 0; i < NTB; ++i) wn[j][i] = ld_nt_f16x8(wbase[i] + (long)kt * 512);
                    xn2[j] = ld_f16x8(p.Xh + (long)crow * p.ldxh + kt * 32 + g * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < GV_CH; ++j) {
                f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                if (base + j < kt1 && rowok) o = xn[j];
                xf[j] = o;
            }
#pragma unroll
            for (int j = 0; j < GV_CH; ++j)
                if (base + j < kt1) {
#pragma unroll
                    for (int i = 0; i < NTB; ++i) acc[i] = mfma16(wf[j][i], xf[j], acc[i]);
                }
            if (more) {
#pragma unroll
                for (int j = 0; j < GV_CH; ++j) {
                    xn[j] = xn2[j];
#pragma unroll
                    for (int i = 0; i < NTB; ++i) wf[j][i] = wn[j][i];
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < GV_CH; ++j)
            if (kt0 + j < kt1) {
#pragma unroll
                for (int i = 0; i < NTB; ++i) acc[i] = mfma16(wf[j][i], xf[j], acc[i]);
            }
    }

    // ---- cross-wave K reduction through LDS (fixed order), epilogue by wave i < NTB for tile i
#pragma unroll
    for (int i = 0; i < NTB; ++i)
        *reinterpret_cast<f32x4*>(accred + (((long)wave * NTB + i) * 64 + lane) * 4) = acc[i];
    WLX_TR_MARK(3);
    __syncthreads();
    if (wave >= NTB) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < nw; ++w) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(accred + (((long)w * NTB + wave) * 64 + lane) * 4);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    WLX_TR_MARK(4);
    if (epi) {
    const float o[4] = {v[0] + bias_e.x, v[1] + bias_e.y, v[2] + bias_e.z, v[3] + bias_e.w};
    const int m = c, n = n_e;
    if constexpr (OUT == GEMV_OUT_F16 || OUT == GEMV_OUT_GELU_F16) {
        float og[4] = {o[0], o[1], o[2], o[3]};
        if constexpr (OUT == GEMV_OUT_GELU_F16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) og[r] = gelu_erf(og[r]);
        }
        const f16x4 h = {(half_t)(og[0] * p.qscale), (half_t)(og[1] * p.qscale),
                         (half_t)(og[2] * p.qscale), (half_t)(og[3] * p.qscale)};   // qscale = 1 unless a q projection
        *reinterpret_cast<f16x4*>(p.Yh + (long)m * p.ldyh + n) = h;
    } else if constexpr (OUT == GEMV_OUT_F32) {
        if (n + 3 < p.N) {
            *reinterpret_cast<float4*>(p.Y + (long)m * p.ldy + n) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.N) p.Y[(long)m * p.ldy + n + r] = o[r];
        }
    } else if constexpr (OUT == GEMV_OUT_RESID) {
        *reinterpret_cast<float4*>(p.Xres + (long)m * p.ldxres + n) =
            make_float4(res_e.x + o[0], res_e.y + o[1], res_e.z + o[2], res_e.w + o[3]);
    } else {   // GEMV_OUT_QKV
        if (n < p.d) {
            const f16x4 h = {(half_t)(o[0] * p.qscale), (half_t)(o[1] * p.qscale),
                             (half_t)(o[2] * p.qscale), (half_t)(o[3] * p.qscale)};
            *reinterpret_cast<f16x4*>(p.Yh + (long)m * p.ldyh + n) = h;
        } else {
            const f16x4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
            int rc = rc_e, rp = rp_e;
            asm volatile("" : "+v"(rc), "+v"(rp));   // keep the address arithmetic (and its wait) down here
            const long kvoff = (long)rc * p.cache_row_stride + (long)rp * p.d;
            if (n < 2 * p.d) *reinterpret_cast<f16x4*>(p.Kc + kvoff + (n - p.d)) = h;
            else *reinterpret_cast<f16x4*>(p.Vc + kvoff + (n - 2 * p.d)) = h;
        }
    }
    }
    WLX_TR_MARK(5);
    WLX_TR_END(p.trc);
}

bool g_decode_v1 = false;
bool g_decode_v2 = false;   // WLX_DECODE_V2=1: second-generation decode kernels (A/B reference)

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
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_addf(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false);
    return v + __builtin_bit_cast(float, t);
}
// sum over the 64 lanes, returned wave-uniform (SGPR)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_addf<0xB1, 0xF>(v);     // quad_perm:[1,0,3,2]
    v = dpp_addf<0x4E, 0xF>(v);     // quad_perm:[2,3,0,1]
    v = dpp_addf<0x141, 0xF>(v);    // row_half_mirror
    v = dpp_addf<0x140, 0xF>(v);    // row_mirror: every lane holds its 16-lane row's sum
    v = dpp_addf<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
    v = dpp_addf<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3: lane 63 holds the wave sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int CH, int LNV, int IN, int OUT, int NTB>
__global__ __launch_bounds__(1024) void dec_gemv2_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    WLX_TR_BEGIN();
    float* accred = smem;                                                  // [nw][NTB][64][4]
    half_t* xs = reinterpret_cast<half_t*>(smem + nw * NTB * 256);         // LN / XATTN: fp16 activation rows

    const int kw0 = wave * p.KTW;                                          // first k-tile of this wave
    const half_t* wp = p.Wp + ((long)(blockIdx.x * NTB) * p.KT + kw0) * 512 + lane * 8;
    const long wstep = (long)p.KT * 512;                                   // next n-tile
    f16x8 wf[CH][NTB];
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int i = 0; i < NTB; ++i) wf[j][i] = ld_nt_f16x8(wp + i * wstep + j * 512);

    // epilogue operands, requested now (every lane, clamped row: no branch around a load)
    const int crow = (c < p.M) ? c : 0;
    const int nt_e = blockIdx.x * NTB + ((wave < NTB) ? wave : 0);
    const int n_e = nt_e * 16 + g * 4;
    float4 bias_e = make_float4(0.f, 0.f, 0.f, 0.f), res_e = bias_e;
    int rc_e = 0, rp_e = 0;
    if constexpr (OUT != GEMV_OUT_F32) bias_e = *reinterpret_cast<const float4*>(p.bias + n_e);
    if constexpr (OUT == GEMV_OUT_RESID) res_e = *reinterpret_cast<const float4*>(p.Xres + (long)crow * p.ldxres + n_e);
    if constexpr (OUT == GEMV_OUT_QKV) { rc_e = p.row_cache[crow]; rp_e = p.row_pos[crow]; }

    f32x4 acc[NTB];
#pragma unroll
    for (int i = 0; i < NTB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 xf[CH];

    if constexpr (IN == GEMV_IN_F16) {
        const half_t* xp = p.Xh + (long)crow * p.ldxh + kw0 * 32 + g * 8;
#pragma unroll
        for (int j = 0; j < CH; ++j) xf[j] = ld_f16x8(xp + j * 32);
        WLX_TR_MARK(1);
#pragma unroll 1
        for (int ch = 1; ch < p.NCH; ++ch) {                               // big-K layers of the larger models only
            f16x8 wn[CH][NTB], xn[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
#pragma unroll
                for (int i = 0; i < NTB; ++i) wn[j][i] = ld_nt_f16x8(wp + i * wstep + (ch * CH + j) * 512);
                xn[j] = ld_f16x8(xp + (ch * CH + j) * 32);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
#pragma unroll
                for (int i = 0; i < NTB; ++i) { acc[i] = mfma16(wf[j][i], xf[j], acc[i]); wf[j][i] = wn[j][i]; }
                xf[j] = xn[j];
            }
        }
    } else if constexpr (IN == GEMV_IN_LN) {
        // wave w normalises rows w, w + nw, ...; a row = LNV float4 per lane (d_model = 256 LNV)
        const float4* g4 = reinterpret_cast<const float4*>(p.gamma) + lane;
        const float4* b4 = reinterpret_cast<const float4*>(p.beta) + lane;
        float4 gq[LNV], bq[LNV];
#pragma unroll
        for (int j = 0; j < LNV; ++j) { gq[j] = g4[64 * j]; bq[j] = b4[64 * j]; }
        const int ldxs = p.K + 8;
        constexpr float invK = 1.0f / (256.0f * LNV);
#pragma unroll 1
        for (int r = wave; r < p.M; r += nw) {
            const float4* x4 = reinterpret_cast<const float4*>(p.X + (long)r * p.ldx) + lane;
            float4 x[LNV];
#pragma unroll
            for (int j = 0; j < LNV; ++j) x[j] = x4[64 * j];
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < LNV; ++j) sm += (x[j].x + x[j].y) + (x[j].z + x[j].w);
            const float mean = wave_sum_dpp(sm) * invK;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < LNV; ++j) {
                x[j].x -= mean; x[j].y -= mean; x[j].z -= mean; x[j].w -= mean;
                q += (x[j].x * x[j].x + x[j].y * x[j].y) + (x[j].z * x[j].z + x[j].w * x[j].w);
            }
            const float rstd = rsqrtf(wave_sum_dpp(q) * invK + 1e-5f);
            half_t* dst = xs + (long)r * ldxs + lane * 4;
#pragma unroll
            for (int j = 0; j < LNV; ++j) {
                const f16x4 hv = {(half_t)(x[j].x * rstd * gq[j].x + bq[j].x), (half_t)(x[j].y * rstd * gq[j].y + bq[j].y),
                                  (half_t)(x[j].z * rstd * gq[j].z + bq[j].z), (half_t)(x[j].w * rstd * gq[j].w + bq[j].w)};
                *reinterpret_cast<f16x4*>(dst + 256 * j) = hv;
            }
        }
        WLX_TR_MARK(1);
        __syncthreads();
        const half_t* xr = xs + c * ldxs + kw0 * 32 + g * 8;                // rows >= M: whatever LDS holds (never stored)
#pragma unroll
        for (int j = 0; j < CH; ++j) xf[j] = *reinterpret_cast<const f16x8*>(xr + j * 32);
    } else {   // GEMV_IN_XATTN: combine the WLX_XSPLIT (m, l, O) partials of the heads in this wave's K slice
        const int ldxs = p.KTW * 32;
        half_t* xw = xs + (long)wave * 16 * ldxs;
        const int nh = p.KTW >> 1, h0 = kw0 >> 1;
        const int n_it = p.M * nh * 16;                                     // (row m, head hh, 4-float group q4)
#pragma unroll 1
        for (int it = lane; it < n_it; it += 64) {
            const int q4 = it & 15, t2 = it >> 4;
            const int m = t2 / nh, hh = t2 - m * nh;
            const int item = m / p.R, qi = m - item * p.R;
            const long pb = (((long)item * p.H + h0 + hh) * WLX_XSPLIT) * 16 + qi;
            const float* mlp = p.part_ml + pb * 2;
            const float* op = p.part_o + pb * 64 + q4 * 4;
            float2 ml[WLX_XSPLIT];
            float4 ov[WLX_XSPLIT];
#pragma unroll
            for (int sp = 0; sp < WLX_XSPLIT; ++sp) {
                ml[sp] = *reinterpret_cast<const float2*>(mlp + sp * 32);
                ov[sp] = *reinterpret_cast<const float4*>(op + sp * 1024);
            }
            float mmax = ml[0].x;
#pragma unroll
            for (int sp = 1; sp < WLX_XSPLIT; ++sp) mmax = fmaxf(mmax, ml[sp].x);
            float den = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f;
#pragma unroll
            for (int sp = 0; sp < WLX_XSPLIT; ++sp) {
                const float w = __expf(ml[sp].x - mmax);
                den += w * ml[sp].y;
                n0 += w * ov[sp].x; n1 += w * ov[sp].y; n2 += w * ov[sp].z; n3 += w * ov[sp].w;
            }
            const float inv = 1.0f / den;
            const f16x4 hv = {(half_t)(n0 * inv), (half_t)(n1 * inv), (half_t)(n2 * inv), (half_t)(n3 * inv)};
            *reinterpret_cast<f16x4*>(xw + m * ldxs + hh * 64 + q4 * 4) = hv;
        }
        WLX_TR_MARK(1);
        // the wave reads back only what it wrote itself: no barrier, the LDS wait is enough
        const half_t* xr = xw + c * ldxs + g * 8;
#pragma unroll
        for (int j = 0; j < CH; ++j) xf[j] = *reinterpret_cast<const f16x8*>(xr + j * 32);
    }
    WLX_TR_MARK(2);
#pragma unroll
    for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int i = 0; i < NTB; ++i) acc[i] = mfma16(wf[j][i], xf[j], acc[i]);

    // ---- cross-wave K reduction through LDS in a fixed order; wave i < NTB finishes tile i
#pragma unroll
    for (int i = 0; i < NTB; ++i) *reinterpret_cast<f32x4*>(accred + ((wave * NTB + i) * 64 + lane) * 4) = acc[i];
    WLX_TR_MARK(3);
    __syncthreads();
    if (wave >= NTB) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* ar = accred + (wave * 64 + lane) * 4;
#pragma unroll 2
    for (int w = 0; w < nw; ++w) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(ar + w * (NTB * 256));
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    WLX_TR_MARK(4);
    if (c < p.M && (NTB == 1 || nt_e * 16 < p.N)) {
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
            *reinterpret_cast<float4*>(p.Xres + (long)c * p.ldxres + n_e) =
                make_float4(res_e.x + o0, res_e.y + o1, res_e.z + o2, res_e.w + o3);
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
    WLX_TR_MARK(5);
    WLX_TR_END(p.trc);
}

struct Gemv2Cfg { bool ok; int nw, CH, NCH, LNV, NTB; size_t shm; };
static Gemv2Cfg gemv2_cfg(const GemvParams& p) {
    Gemv2Cfg c{};
    c.ok = false;
    if (g_decode_v1 || g_decode_v2 || p.M > 16 || p.M < 1) return c;
    if (p.bias ? (p.N & 15) != 0 : p.out_mode != GEMV_OUT_F32) return c;      // bias <=> not the vocabulary projection
    const bool combo = (p.in_mode == GEMV_IN_LN && (p.out_mode == GEMV_OUT_QKV || p.out_mode == GEMV_OUT_F16 ||
                                                    p.out_mode == GEMV_OUT_GELU_F16 || p.out_mode == GEMV_OUT_F32)) ||
                       (p.in_mode != GEMV_IN_LN && p.out_mode == GEMV_OUT_RESID);
    if (!combo || p.K != p.KT * 32) return c;
    const int cap = (p.in_mode == GEMV_IN_F16) ? 16 : 8;
    // exact factorisation KT = nw * CH * NCH, CH in {6, 5, 4}: fewest chunks first, then the widest chunk
    int best_nch = 1 << 30;
    for (int CH = 6; CH >= 4; --CH) {
        if (p.KT % CH) continue;
        const int q = p.KT / CH;                    // = nw * NCH
        for (int nw = std::min(cap, q); nw >= 1; --nw) {
            if (q % nw) continue;
            const int nch = q / nw;
            if (nch < best_nch) { best_nch = nch; c.nw = nw; c.CH = CH; c.NCH = nch; }
            break;
        }
    }
    if (best_nch == (1 << 30)) return c;
    if (p.in_mode != GEMV_IN_F16 && c.NCH != 1) return c;
    if (p.in_mode == GEMV_IN_XATTN && ((c.CH * c.NCH) & 1)) return c;        // a wave's K slice holds whole heads
    c.LNV = 0;
    if (p.in_mode == GEMV_IN_LN) {
        if (p.K % 256 || p.K / 256 < 2 || p.K / 256 > 5) return c;
        c.LNV = p.K / 256;
    }
    c.NTB = (p.out_mode == GEMV_OUT_F32 && p.N > 8192) ? 2 : 1;
    c.shm = sizeof(float) * (size_t)c.nw * c.NTB * 256;
    if (p.in_mode == GEMV_IN_LN) c.shm += (size_t)16 * (p.K + 8) * sizeof(half_t);
    if (p.in_mode == GEMV_IN_XATTN) c.shm += (size_t)c.nw * 16 * (c.CH * c.NCH * 32) * sizeof(half_t);
    c.ok = true;
    return c;
}

template <int CH, int LNV>
static bool gemv2_launch_ln(const GemvParams& p, const Gemv2Cfg& c, dim3 grid, dim3 block, hipStream_t s) {
    switch (p.out_mode) {
        case GEMV_OUT_QKV: hipLaunchKernelGGL((dec_gemv2_kernel<CH, LNV, GEMV_IN_LN, GEMV_OUT_QKV, 1>), grid, block, c.shm, s, p); return true;
        case GEMV_OUT_F16: hipLaunchKernelGGL((dec_gemv2_kernel<CH, LNV, GEMV_IN_LN, GEMV_OUT_F16, 1>), grid, block, c.shm, s, p); return true;
        case GEMV_OUT_GELU_F16: hipLaunchKernelGGL((dec_gemv2_kernel<CH, LNV, GEMV_IN_LN, GEMV_OUT_GELU_F16, 1>), grid, block, c.shm, s, p); return true;
        case GEMV_OUT_F32:
            if (c.NTB == 2) hipLaunchKernelGGL((dec_gemv2_kernel<CH, LNV, GEMV_IN_LN, GEMV_OUT_F32, 2>), grid, block, c.shm, s, p);
            else hipLaunchKernelGGL((dec_gemv2_kernel<CH, LNV, GEMV_IN_LN, GEMV_OUT_F32, 1>), grid, block, c.shm, s, p);
            return true;
        default: return false;
    }
}
template <int CH>
static bool gemv2_launch_other(const GemvParams& p, const Gemv2Cfg& c, dim3 grid, dim3 block, hipStream_t s) {
    if (p.in_mode == GEMV_IN_F16) hipLaunchKernelGGL((dec_gemv2_kernel<CH, 1, GEMV_IN_F16, GEMV_OUT_RESID, 1>), grid, block, c.shm, s, p);
    else hipLaunchKernelGGL((dec_gemv2_kernel<CH, 1, GEMV_IN_XATTN, GEMV_OUT_RESID, 1>), grid, block, c.shm, s, p);
    return true;
}
// the (CH, LNV) pairs of the Whisper family: d_model 512 (4,2), 768 (6,3), 1024 (4,4), 1280 (5,5)
static bool gemv2_launch(const GemvParams& p0, const Gemv2Cfg& c, hipStream_t s) {
    GemvParams p = p0;
    p.KTW = c.CH * c.NCH; p.NCH = c.NCH;
#ifdef WLX_TRACE
    { static thread_local char nm[512][48]; const int q = g_trace_seq < 512 ? g_trace_seq : 511;
      snprintf(nm[q], 48, "gemv2<%d,%d> N%d K%d", p.in_mode, p.out_mode, p.N, p.K); p.trc = trace_next(nm[q]); }
#endif
    const int NT_total = (p.N + 15) / 16;
    dim3 grid((NT_total + c.NTB - 1) / c.NTB), block(c.nw * 64);
    if (p.in_mode == GEMV_IN_LN) {
        if (c.CH == 6 && c.LNV == 3) return gemv2_launch_ln<6, 3>(p, c, grid, block, s);
        if (c.CH == 5 && c.LNV == 5) return gemv2_launch_ln<5, 5>(p, c, grid, block, s);
        if (c.CH == 4 && c.LNV == 2) return gemv2_launch_ln<4, 2>(p, c, grid, block, s);
        if (c.CH == 4 && c.LNV == 4) return gemv2_launch_ln<4, 4>(p, c, grid, block, s);
        return false;
    }
    switch (c.CH) {
        case 6: return gemv2_launch_other<6>(p, c, grid, block, s);
        case 5: return gemv2_launch_other<5>(p, c, grid, block, s);
        default: return gemv2_launch_other<4>(p, c, grid, block, s);
    }
}
static bool gemv2_ok(const GemvParams& p, Gemv2Cfg* out) {
    const Gemv2Cfg c = gemv2_cfg(p);
    if (!c.ok) return false;
    if (p.in_mode == GEMV_IN_LN) {
        const bool pair = (c.CH == 6 && c.LNV == 3) || (c.CH == 5 && c.LNV == 5) || (c.CH == 4 && c.LNV == 2) || (c.CH == 4 && c.LNV == 4);
        if (!pair) return false;
    }
    if (out) *out = c;
    return true;
}

struct Gemv1Cfg { int nw, KTW, NTB, maxt; size_t shm; };
static Gemv1Cfg gemv1_cfg(const GemvParams& p) {
    Gemv1Cfg c;
    const int cap = (p.in_mode == GEMV_IN_LN) ? 8 : (p.in_mode == GEMV_IN_XATTN) ? 12 : 16;
    int nw = (p.KT + GV_CH - 1) / GV_CH;
    // cross-attention combine: one head (2 k-tiles) per wave when they fit, so the M x 16 x WLX_XSPLIT partial loads of
    // a head are ONE round of <= 2 items per lane instead of several dependent rounds
    if (p.in_mode == GEMV_IN_XATTN) nw = p.KT / 2;
    if (nw < 1) nw = 1;
    if (nw > cap) nw = cap;
    int KTW = 2 * ((p.KT + 2 * nw - 1) / (2 * nw));     // even: a wave's K slice holds whole heads
    nw = (p.KT + KTW - 1) / KTW;
    c.nw = nw; c.KTW = KTW;
    c.NTB = (p.out_mode == GEMV_OUT_F32 && p.N > 8192 && nw >= 2) ? 2 : 1;
    c.maxt = (p.in_mode == GEMV_IN_LN) ? 512 : (p.in_mode == GEMV_IN_XATTN) ? 768 : 1024;
    c.shm = sizeof(float) * ((size_t)2 * nw * 16 + (size_t)nw * c.NTB * 256);
    if (p.in_mode == GEMV_IN_XATTN) c.shm += (size_t)nw * 16 * KTW * 32 * sizeof(half_t);
    if (p.in_mode == GEMV_IN_LN) c.shm += (size_t)16 * (p.K + 8) * sizeof(half_t);          // fp16 LN(x) rows
    return c;
}
static bool gemv1_ok(const GemvParams& p) {
    if (g_decode_v1 || p.M > 16) return false;
    if (p.bias ? (p.N & 15) != 0 : p.out_mode != GEMV_OUT_F32) return false;   // bias <=> not the vocabulary projection
    const bool combo = (p.in_mode == GEMV_IN_LN && (p.out_mode == GEMV_OUT_QKV || p.out_mode == GEMV_OUT_F16 ||
                                                    p.out_mode == GEMV_OUT_GELU_F16 || p.out_mode == GEMV_OUT_F32)) ||
                       (p.in_mode != GEMV_IN_LN && p.out_mode == GEMV_OUT_RESID);
    if (!combo) return false;
    if (p.in_mode != GEMV_IN_F16 && (p.KT + GV_CH - 1) / GV_CH > 8) return false;
    if (p.in_mode != GEMV_IN_F16 && gemv1_cfg(p).KTW > GV_CH) return false;
    return true;
}

const char* dec_gemv_kernel_name(const GemvParams& p) {
    static thread_local char buf[64];
    const int MT = (p.M + 15) / 16;
    Gemv2Cfg c2;
    if (gemv2_ok(p, &c2)) {
        snprintf(buf, sizeof(buf), "dec_gemv2_kernel<%d, %d, %d, %d, %d>", c2.CH, c2.LNV ? c2.LNV : 1, p.in_mode, p.out_mode, c2.NTB);
        return buf;
    }
    if (gemv1_ok(p)) {
        const Gemv1Cfg c = gemv1_cfg(p);
        snprintf(buf, sizeof(buf), "dec_gemv1_kernel<%d, %d, %d, %d>", c.NTB, p.in_mode, p.out_mode, c.maxt);
    } else {
        snprintf(buf, sizeof(buf), "dec_gemv_kernel<%d, %d, %d>", MT > 4 ? 4 : MT, MT == 1 ? 2 : 1, p.in_mode);
    }
    return buf;
}

template <int MT, int NTB>
static void gemv_dispatch_in(const GemvParams& p, dim3 grid, dim3 block, size_t shm, hipStream_t s) {
    switch (p.in_mode) {
        case GEMV_IN_LN: hipLaunchKernelGGL((dec_gemv_kernel<MT, NTB, GEMV_IN_LN>), grid, block, shm, s, p); break;
        case GEMV_IN_F16: hipLaunchKernelGGL((dec_gemv_kernel<MT, NTB, GEMV_IN_F16>), grid, block, shm, s, p); break;
        default: hipLaunchKernelGGL((dec_gemv_kernel<MT, NTB, GEMV_IN_XATTN>), grid, block, shm, s, p); break;
    }
}

void launch_dec_gemv(const GemvParams& p0, hipStream_t s) {
    Gemv2Cfg c2;
    if (gemv2_ok(p0, &c2) && gemv2_launch(p0, c2, s)) return;
    if (gemv1_ok(p0)) {
        GemvParams p = p0;
        const Gemv1Cfg c = gemv1_cfg(p);
        p.KTW = c.KTW;
#ifdef WLX_TRACE
        { static thread_local char nm[512][48]; const int q = g_trace_seq < 512 ? g_trace_seq : 511;
          snprintf(nm[q], 48, "gemv1<%d,%d> N%d K%d", p.in_mode, p.out_mode, p.N, p.K); p.trc = trace_next(nm[q]); }
#endif
        const int NT_total = (p.N + 15) / 16;
        dim3 grid((NT_total + c.NTB - 1) / c.NTB), block(c.nw * 64);
#define WLX_G1(NTB_, IN_, OUT_, MAXT_) \
    hipLaunchKernelGGL((dec_gemv1_kernel<NTB_, IN_, OUT_, MAXT_>), grid, block, c.shm, s, p)
        if (p.in_mode == GEMV_IN_LN) {
            switch (p.out_mode) {
                case GEMV_OUT_QKV: WLX_G1(1, GEMV_IN_LN, GEMV_OUT_QKV, 512); break;
                case GEMV_OUT_F16: WLX_G1(1, GEMV_IN_LN, GEMV_OUT_F16, 512); break;
                case GEMV_OUT_GELU_F16: WLX_G1(1, GEMV_IN_LN, GEMV_OUT_GELU_F16, 512); break;
                default:
                    if (c.NTB == 2) WLX_G1(2, GEMV_IN_LN, GEMV_OUT_F32, 512);
                    else WLX_G1(1, GEMV_IN_LN, GEMV_OUT_F32, 512);
                    break;
            }
        } else if (p.in_mode == GEMV_IN_F16) {
            WLX_G1(1, GEMV_IN_F16, GEMV_OUT_RESID, 1024);
        } else {
            WLX_G1(1, GEMV_IN_XATTN, GEMV_OUT_RESID, 768);
        }
#undef WLX_G1
        return;
    }
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
// one wave per (row, head); positions 0..pos[row]; history of the row through the ancestry table
__global__ __launch_bounds__(64) void dec_self_attn_kernel(const half_t* __restrict__ q, long ldq,
                                                           const half_t* __restrict__ Kc,
                                                           const half_t* __restrict__ Vc, long crs, int d,
                                                           const int* __restrict__ pos,
                                                           const int* __restrict__ ancrow,
                                                           const short* __restrict__ anc,
                                                           half_t* __restrict__ out, long ldo,
                                                           const int* __restrict__ done) {
    if (done && *done) return;
    __shared__ float prob[WLX_T_TEXT];
    __shared__ int crow[WLX_T_TEXT];
    const int lane = threadIdx.x;
    const int r = blockIdx.x, h = blockIdx.y;
    const int len = pos[r] + 1;
    const short* ar = anc + (long)ancrow[r] * WLX_T_TEXT;

    // q (already scaled) -> registers as 64 floats? keep as 8 x f16x8 broadcast loads
    f16x8 qv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = ld_f16x8(q + (long)r * ldq + h * WLX_HEAD_DIM + i * 8);

    float lmax = WLX_NEG_INF;
    for (int p0 = 0; p0 < len; p0 += 64) {
        const int p = p0 + lane;
        float sc = WLX_NEG_INF;
        if (p < len) {
            const int cr = ar[p];
            crow[p] = cr;
            const half_t* kp = Kc + (long)cr * crs + (long)p * d + h * WLX_HEAD_DIM;
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f16x8 kv = ld_f16x8(kp + i * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) a = fmaf((float)kv[e], (float)qv[i][e], a);
            }
            sc = a;
            prob[p] = a;
        }
        lmax = fmaxf(lmax, sc);
    }
    lmax = wave_max(lmax);
    float lsum = 0.f;
    for (int p = lane; p < len; p += 64) {
        float e = __expf(prob[p] - lmax);
        prob[p] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    // out[dd = lane] = sum_p prob[p] * V[p][dd]
    float o = 0.f;
    const half_t* vb = Vc + h * WLX_HEAD_DIM + lane;
    int p = 0;
    for (; p + 4 <= len; p += 4) {
        float v0 = (float)vb[(long)crow[p] * crs + (long)p * d];
        float v1 = (float)vb[(long)crow[p + 1] * crs + (long)(p + 1) * d];
        float v2 = (float)vb[(long)crow[p + 2] * crs + (long)(p + 2) * d];
        float v3 = (float)vb[(long)crow[p + 3] * crs + (long)(p + 3) * d];
        o = fmaf(prob[p], v0, o); o = fmaf(prob[p + 1], v1, o);
        o = fmaf(prob[p + 2], v2, o); o = fmaf(prob[p + 3], v3, o);
    }
    for (; p < len; ++p) o = fmaf(prob[p], (float)vb[(long)crow[p] * crs + (long)p * d], o);
    out[(long)r * ldo + h * WLX_HEAD_DIM + lane] = (half_t)(o / lsum);
}

// second generation: same arithmetic for the scores; the V pass gives every lane a (position group, 8-dim
// chunk) pair so 8 independent 16-byte loads are in flight per lane (one L2 round trip per 64 positions
// instead of one per 4), and the done flag / row tables / q are requested before the first wait.
__global__ __launch_bounds__(64) void dec_self_attn2_kernel(const half_t* __restrict__ q, long ldq,
                                                            const half_t* __restrict__ Kc,
                                                            const half_t* __restrict__ Vc, long crs, int d,
                                                            const int* __restrict__ pos,
                                                            const int* __restrict__ ancrow,
                                                            const short* __restrict__ anc,
                                                            half_t* __restrict__ out, long ldo,
                                                            const int* __restrict__ done WLX_TR_PARAM) {
    __shared__ float prob[WLX_T_TEXT];
    __shared__ int crow[WLX_T_TEXT];
    const int lane = threadIdx.x;
    const int r = blockIdx.x, h = blockIdx.y;
    WLX_TR_BEGIN();
    int dn = 0;
    if (done) dn = *done;
    const int len = pos[r] + 1;
    const short* ar = anc + (long)ancrow[r] * WLX_T_TEXT;
    f16x8 qv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = ld_f16x8(q + (long)r * ldq + h * WLX_HEAD_DIM + i * 8);
    if (dn) return;
    WLX_TR_MARK(1);

    float lmax = WLX_NEG_INF;
    for (int p0 = 0; p0 < len; p0 += 64) {
        const int p = p0 + lane;
        float sc = WLX_NEG_INF;
        if (p < len) {
            const int cr = ar[p];
            crow[p] = cr;
            const half_t* kp = Kc + (long)cr * crs + (long)p * d + h * WLX_HEAD_DIM;
            f16x8 kv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) kv[i] = ld_f16x8(kp + i * 8);
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a = fmaf((float)kv[i][e], (float)qv[i][e], a);
            }
            sc = a;
            prob[p] = a;
        }
        lmax = fmaxf(lmax, sc);
    }
    lmax = wave_max(lmax);
    float lsum = 0.f;
    for (int p = lane; p < len; p += 64) {
        const float e = __expf(prob[p] - lmax);
        prob[p] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    WLX_TR_MARK(2);
    // out[dd] = sum_p prob[p] * V[p][dd]; lane = (pg, dc): positions p = pg (mod 8), dims dc*8 .. dc*8+7
    const int pg = lane >> 3, dc = lane & 7;
    const half_t* vb = Vc + h * WLX_HEAD_DIM + dc * 8;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < len; p0 += 64) {
        f16x8 vv[8];
        float pr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * 8 + pg;
            const int pc = (p < len) ? p : len - 1;
            pr[u] = (p < len) ? prob[pc] : 0.f;
            vv[u] = ld_f16x8(vb + (long)crow[pc] * crs + (long)pc * d);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pr[u], (float)vv[u][e], o[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] += __shfl_xor(o[e], 8, 64);
        o[e] += __shfl_xor(o[e], 16, 64);
        o[e] += __shfl_xor(o[e], 32, 64);
    }
    if (pg == 0) {
        const float inv = 1.0f / lsum;
        const f16x8 hv = {(half_t)(o[0] * inv), (half_t)(o[1] * inv), (half_t)(o[2] * inv), (half_t)(o[3] * inv),
                          (half_t)(o[4] * inv), (half_t)(o[5] * inv), (half_t)(o[6] * inv), (half_t)(o[7] * inv)};
        *reinterpret_cast<f16x8*>(out + (long)r * ldo + h * WLX_HEAD_DIM + dc * 8) = hv;
    }
    WLX_TR_MARK(3);
    WLX_TR_END(trc);
}

void launch_dec_self_attn(const half_t* q, long ldq, const half_t* Kc, const half_t* Vc, long crs, int d, int H,
                          const RowTables& rt, int rows, half_t* out, long ldo, const int* done, hipStream_t s) {
    if (g_decode_v1)
        hipLaunchKernelGGL(dec_self_attn_kernel, dim3(rows, H), dim3(64), 0, s, q, ldq, Kc, Vc, crs, d, rt.pos,
                           rt.ancrow, rt.anc, out, ldo, done);
    else
        hipLaunchKernelGGL(dec_self_attn2_kernel, dim3(rows, H), dim3(64), 0, s, q, ldq, Kc, Vc, crs, d, rt.pos,
                           rt.ancrow, rt.anc, out, ldo, done WLX_TR_ARG("self_attn2"));
}

// ------------------------------------------------------------------ decode cross-attention (flash-decoding split over keys)
// grid (split, head, item); the R rows of an item share the item's encoder K/V, so they form ONE
// 16-row MFMA query tile (same transposed-score scheme as attention.hip). Each split writes its
// un-normalised (m, l, O) partial; the consumer projection combines them in its prologue.
__global__ __launch_bounds__(64) void dec_cross_attn_kernel(const half_t* __restrict__ q, long ldq,
                                                            const half_t* __restrict__ Kx, long ldk, long isk,
                                                            const half_t* __restrict__ Vtx, long ldvt, long isv,
                                                            int H, int R, int rows,
                                                            const int* __restrict__ group_item,
                                                            float* __restrict__ part_o, float* __restrict__ part_ml,
                                                            const int* __restrict__ done) {
    if (done && *done) return;
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int sp = blockIdx.x, h = blockIdx.y, grp = blockIdx.z;
    const int item = group_item[grp];
    constexpr int T = WLX_T_AUDIO;
    constexpr int TILES = (T + 31) / 32;                               // 47
    constexpr int TPS = (TILES + WLX_XSPLIT - 1) / WLX_XSPLIT;         // 6 key tiles per split
    const int tile0 = sp * TPS;
    const int tile1 = (tile0 + TPS < TILES) ? tile0 + TPS : TILES;

    const half_t* K = Kx + (long)item * isk + h * WLX_HEAD_DIM;
    const half_t* Vt = Vtx + (long)item * isv + (long)h * WLX_HEAD_DIM * ldvt;

    int row = grp * R + c;
    const bool qok = (c < R) && (row < rows);
    if (!qok) row = grp * R;  // any valid row; result discarded
    f16x8 qf[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) qf[kt] = ld_f16x8(q + (long)row * ldq + h * WLX_HEAD_DIM + kt * 32 + g * 8);

    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = WLX_NEG_INF, lrun = 0.f;
    const half_t* kbase = K + (long)c * ldk + g * 8;
    const half_t* vbase = Vt + (long)c * ldvt + g * 4;

    for (int tile = tile0; tile < tile1; ++tile) {
        const int key0 = tile * 32;
        f16x8 kf[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kf[s2][kt] = ld_f16x8(kbase + (long)(key0 + s2 * 16) * ldk + kt * 32);
        f16x8 vf[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const half_t* vp = vbase + (long)dt * 16 * ldvt + key0;
            f16x4 lo = ld_f16x4(vp), hi = ld_f16x4(vp + 16);
            vf[dt] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
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
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { pv[i] = __expf(pv[i] - mnew); psum += pv[i]; }
        lrun = lrun * alpha + psum;
        mrun = mnew;
        f16x8 pf = {(half_t)pv[0], (half_t)pv[1], (half_t)pv[2], (half_t)pv[3],
                    (half_t)pv[4], (half_t)pv[5], (half_t)pv[6], (half_t)pv[7]};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 a = acc[dt];
            a[0] *= alpha; a[1] *= alpha; a[2] *= alpha; a[3] *= alpha;
            acc[dt] = mfma16(vf[dt], pf, a);
        }
    }
    lrun += __shfl_xor(lrun, 16, 64);
    lrun += __shfl_xor(lrun, 32, 64);
    const long pb = (((long)grp * H + h) * WLX_XSPLIT + sp) * 16 + c;
    if (g == 0) *reinterpret_cast<float2*>(part_ml + pb * 2) = make_float2(mrun, lrun);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<f32x4*>(part_o + pb * 64 + dt * 16 + g * 4) = acc[dt];
}

// second generation: identical arithmetic and partial layout, but the K and V^T fragments of all six key
// tiles of the split (48 KiB per wave) are requested in one burst before the first MFMA, so the split costs one
// HBM/L2 round trip instead of six dependent ones. K/V^T are padded to 1536 keys, so every split may read its full
// six tiles; keys >= 1500 are masked to -inf exactly as before.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void dec_cross_attn2_kernel(const half_t* __restrict__ q, long ldq,
                                                             const half_t* __restrict__ Kx, long ldk, long isk,
                                                             const half_t* __restrict__ Vtx, long ldvt, long isv,
                                                             int H, int R, int rows,
                                                             const int* __restrict__ group_item,
                                                             float* __restrict__ part_o, float* __restrict__ part_ml,
                                                             const int* __restrict__ done WLX_TR_PARAM) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int sp = blockIdx.x, h = blockIdx.y, grp = blockIdx.z;
    WLX_TR_BEGIN();
    int dn = 0;
    if (done) dn = *done;
    const int item = group_item[grp];
    constexpr int T = WLX_T_AUDIO;
    constexpr int TPS = WLX_T_AUDIO_PAD / 32 / WLX_XSPLIT;     // 6 key tiles per split (48 tiles of 32 keys)
    static_assert(TPS * WLX_XSPLIT * 32 == WLX_T_AUDIO_PAD, "key padding must cover every split");
    const int tile0 = sp * TPS;

    const half_t* K = Kx + (long)item * isk + h * WLX_HEAD_DIM;
    const half_t* Vt = Vtx + (long)item * isv + (long)h * WLX_HEAD_DIM * ldvt;
    const half_t* kbase = K + (long)c * ldk + g * 8;
    const half_t* vbase = Vt + (long)c * ldvt + g * 4;

    f16x8 kf[TPS][2][2];
    f16x4 vlo[TPS][4], vhi[TPS][4];
#pragma unroll
    for (int t = 0; t < TPS; ++t) {
        const int key0 = (tile0 + t) * 32;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kf[t][s2][kt] = ld_f16x8(kbase + (long)(key0 + s2 * 16) * ldk + kt * 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const half_t* vp = vbase + (long)dt * 16 * ldvt + key0;
            vlo[t][dt] = ld_f16x4(vp);
            vhi[t][dt] = ld_f16x4(vp + 16);
        }
    }
    int row = grp * R + c;
    const bool qok = (c < R) && (row < rows);
    if (!qok) row = grp * R;  // any valid row; result discarded
    f16x8 qf[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) qf[kt] = ld_f16x8(q + (long)row * ldq + h * WLX_HEAD_DIM + kt * 32 + g * 8);
    if (dn) return;
    WLX_TR_MARK(1);

    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = WLX_NEG_INF, lrun = 0.f;
#pragma unroll
    for (int t = 0; t < TPS; ++t) {
        const int key0 = (tile0 + t) * 32;
        f32x4 st[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            st[s2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) st[s2] = mfma16(kf[t][s2][kt], qf[kt], st[s2]);
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
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);      // finite from the first tile on: every split starts below key 1500
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { pv[i] = __expf(pv[i] - mnew); psum += pv[i]; }
        lrun = lrun * alpha + psum;
        mrun = mnew;
        const f16x8 pf = {(half_t)pv[0], (half_t)pv[1], (half_t)pv[2], (half_t)pv[3],
                          (half_t)pv[4], (half_t)pv[5], (half_t)pv[6], (half_t)pv[7]};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 a = acc[dt];
            a[0] *= alpha; a[1] *= alpha; a[2] *= alpha; a[3] *= alpha;
            const f16x8 vf = {vlo[t][dt][0], vlo[t][dt][1], vlo[t][dt][2], vlo[t][dt][3],
                              vhi[t][dt][0], vhi[t][dt][1], vhi[t][dt][2], vhi[t][dt][3]};
            acc[dt] = mfma16(vf, pf, a);
        }
    }
    lrun += __shfl_xor(lrun, 16, 64);
    lrun += __shfl_xor(lrun, 32, 64);
    const long pb = (((long)grp * H + h) * WLX_XSPLIT + sp) * 16 + c;
    if (g == 0) *reinterpret_cast<float2*>(part_ml + pb * 2) = make_float2(mrun, lrun);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<f32x4*>(part_o + pb * 64 + dt * 16 + g * 4) = acc[dt];
    WLX_TR_MARK(3);
    WLX_TR_END(trc);
}

// third generation: one WORKGROUP of TPS waves per (split, head, group), one 32-key tile per wave. A lone wave pulling
// its split's 48 KiB was the longest single-wave stream of the step (5.6 us per launch alone); six waves pull 8 KiB each,
// form their (m, l, O) over one tile and the workgroup merges the six through LDS into the same per-split partial the
// consumer projection already combines (layout unchanged). Fully masked tiles (keys >= 1500) contribute (m=-inf, l=0, O=0).
#define XA3_TPS (WLX_T_AUDIO_PAD / 32 / WLX_XSPLIT)
__global__ __launch_bounds__(XA3_TPS * 64) void dec_cross_attn3_kernel(const half_t* __restrict__ q, long ldq,
                                                                       const half_t* __restrict__ Kx, long ldk, long isk,
                                                                       const half_t* __restrict__ Vtx, long ldvt, long isv,
                                                                       int H, int R, int rows,
                                                                       const int* __restrict__ group_item,
                                                                       float* __restrict__ part_o, float* __restrict__ part_ml WLX_TR_PARAM) {
    constexpr int TPS = XA3_TPS;
    static_assert(TPS * WLX_XSPLIT * 32 == WLX_T_AUDIO_PAD && TPS >= 4, "key padding must cover every split; >= 4 waves combine");
    __shared__ __attribute__((aligned(16))) float Os[TPS][16][68];
    __shared__ float MLs[TPS][16][2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int sp = blockIdx.x, h = blockIdx.y, grp = blockIdx.z;
    WLX_TR_BEGIN();
    const int item = group_item[grp];
    constexpr int T = WLX_T_AUDIO;
    const int key0 = (sp * TPS + wave) * 32;
    const half_t* K = Kx + (long)item * isk + h * WLX_HEAD_DIM;
    const half_t* Vt = Vtx + (long)item * isv + (long)h * WLX_HEAD_DIM * ldvt;
    const half_t* kbase = K + (long)c * ldk + g * 8;
    const half_t* vbase = Vt + (long)c * ldvt + g * 4;
    f16x8 kf[2][2];
    f16x4 vlo[4], vhi[4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) kf[s2][kt] = ld_f16x8(kbase + (long)(key0 + s2 * 16) * ldk + kt * 32);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const half_t* vp = vbase + (long)dt * 16 * ldvt + key0;
        vlo[dt] = ld_f16x4(vp);
        vhi[dt] = ld_f16x4(vp + 16);
    }
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
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float msafe = (tmax == WLX_NEG_INF) ? 0.f : tmax;      // a fully masked tile: every p = exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { pv[i] = __expf(pv[i] - msafe); psum += pv[i]; }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    const f16x8 pf = {(half_t)pv[0], (half_t)pv[1], (half_t)pv[2], (half_t)pv[3],
                      (half_t)pv[4], (half_t)pv[5], (half_t)pv[6], (half_t)pv[7]};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const f16x8 vf = {vlo[dt][0], vlo[dt][1], vlo[dt][2], vlo[dt][3], vhi[dt][0], vhi[dt][1], vhi[dt][2], vhi[dt][3]};
        const f32x4 a = mfma16(vf, pf, (f32x4){0.f, 0.f, 0.f, 0.f});
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
        const long pb = (((long)grp * H + h) * WLX_XSPLIT + sp) * 16 + c;
        if (dt == 0 && g == 0) *reinterpret_cast<float2*>(part_ml + pb * 2) = make_float2(M, l);
        *reinterpret_cast<f32x4*>(part_o + pb * 64 + dt * 16 + g * 4) = o;
    }
    WLX_TR_MARK(3);
    WLX_TR_END(trc);
}

void launch_dec_cross_attn(const half_t* q, long ldq, const half_t* Kx, long ldk, long isk, const half_t* Vtx,
                           long ldvt, long isv, int H, int R, int groups, int rows, const int* group_item,
                           float* part_o, float* part_ml, const int* done, hipStream_t s) {
    if (g_decode_v1)
        hipLaunchKernelGGL(dec_cross_attn_kernel, dim3(WLX_XSPLIT, H, groups), dim3(64), 0, s, q, ldq, Kx, ldk, isk, Vtx,
                           ldvt, isv, H, R, rows, group_item, part_o, part_ml, done);
    else if (g_decode_v2)
        hipLaunchKernelGGL(dec_cross_attn2_kernel, dim3(WLX_XSPLIT, H, groups), dim3(64), 0, s, q, ldq, Kx, ldk, isk, Vtx,
                           ldvt, isv, H, R, rows, group_item, part_o, part_ml, done WLX_TR_ARG("cross_attn2"));
    else
        hipLaunchKernelGGL(dec_cross_attn3_kernel, dim3(WLX_XSPLIT, H, groups), dim3(XA3_TPS * 64), 0, s, q, ldq, Kx, ldk, isk, Vtx,
                           ldvt, isv, H, R, rows, group_item, part_o, part_ml WLX_TR_ARG("cross_attn3"));
}

}  // namespace wlx
