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
                                                        float* __restrict__ x, const int* __restrict__ done) {
    if (done && *done) return;
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
}

void launch_dec_embed(const half_t* tok_emb, const float* pos_emb, int d, const RowTables& rt, int rows,
                      float* x, const int* done, hipStream_t s) {
    hipLaunchKernelGGL(dec_embed_kernel, dim3(rows), dim3(256), 0, s, tok_emb, pos_emb, d, rt.token, rt.pos,
                       rt.cache, rt.intok, x, done);
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
        // LayerNorm over K = d_model (one chunk per wave, host guarantees KTW <= GV_CH): x, gamma, beta all
        // requested before the statistics are reduced.
        float xr[GV_CH][8], gg[GV_CH][8], bb[GV_CH][8];
#pragma unroll
        for (int j = 0; j < GV_CH; ++j) {
            int kt = kt0 + j;
            if (kt > KT - 1) kt = KT - 1;
            const float4* xp = reinterpret_cast<const float4*>(p.X + (long)crow * p.ldx + kt * 32 + g * 8);
            const float4* gp = reinterpret_cast<const float4*>(p.gamma + kt * 32 + g * 8);
            const float4* bp = reinterpret_cast<const float4*>(p.beta + kt * 32 + g * 8);
            const float4 a = xp[0], b = xp[1], g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
            xr[j][0] = a.x; xr[j][1] = a.y; xr[j][2] = a.z; xr[j][3] = a.w;
            xr[j][4] = b.x; xr[j][5] = b.y; xr[j][6] = b.z; xr[j][7] = b.w;
            gg[j][0] = g0.x; gg[j][1] = g0.y; gg[j][2] = g0.z; gg[j][3] = g0.w;
            gg[j][4] = g1.x; gg[j][5] = g1.y; gg[j][6] = g1.z; gg[j][7] = g1.w;
            bb[j][0] = b0.x; bb[j][1] = b0.y; bb[j][2] = b0.z; bb[j][3] = b0.w;
            bb[j][4] = b1.x; bb[j][5] = b1.y; bb[j][6] = b1.z; bb[j][7] = b1.w;
        }
        if (dn) return;
        const float invK = 1.0f / (float)p.K;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < GV_CH; ++j)
            if (kt0 + j < kt1 && rowok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s += xr[j][e];
            }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0) red[wave * 16 + c] = s;
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < nw; ++w) tot += red[w * 16 + c];
        const float mean = tot * invK;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < GV_CH; ++j)
            if (kt0 + j < kt1 && rowok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dlt = xr[j][e] - mean; q += dlt * dlt; }
            }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (g == 0) red[nw * 16 + wave * 16 + c] = q;
        __syncthreads();
        float qt = 0.f;
        for (int w = 0; w < nw; ++w) qt += red[nw * 16 + w * 16 + c];
        const float rstd = rsqrtf(qt * invK + 1e-5f);
#pragma unroll
        for (int j = 0; j < GV_CH; ++j) {
            f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kt0 + j < kt1 && rowok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)((xr[j][e] - mean) * rstd * gg[j][e] + bb[j][e]);
            }
            xf[j] = o;
        }
    } else if constexpr (IN == GEMV_IN_XATTN) {
        // wave-local combine of the WLX_XSPLIT (m, l, O) partials of the heads inside this wave's K slice
        // (KTW even => whole heads), written as fp16 rows to this wave's LDS image, then read back as B fragments.
        if (dn) return;
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
    __syncthreads();
    if (wave >= NTB) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < nw; ++w) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(accred + (((long)w * NTB + wave) * 64 + lane) * 4);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    if (!epi) return;
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

bool g_decode_v1 = false;

struct Gemv1Cfg { int nw, KTW, NTB, maxt; size_t shm; };
static Gemv1Cfg gemv1_cfg(const GemvParams& p) {
    Gemv1Cfg c;
    const int cap = (p.in_mode == GEMV_IN_F16) ? 16 : 8;
    int nw = (p.KT + GV_CH - 1) / GV_CH;
    if (nw < 1) nw = 1;
    if (nw > cap) nw = cap;
    int KTW = 2 * ((p.KT + 2 * nw - 1) / (2 * nw));     // even: a wave's K slice holds whole heads
    nw = (p.KT + KTW - 1) / KTW;
    c.nw = nw; c.KTW = KTW;
    c.NTB = (p.out_mode == GEMV_OUT_F32 && p.N > 8192 && nw >= 2) ? 2 : 1;
    c.maxt = (p.in_mode == GEMV_IN_F16) ? 1024 : 512;
    c.shm = sizeof(float) * ((size_t)2 * nw * 16 + (size_t)nw * c.NTB * 256);
    if (p.in_mode == GEMV_IN_XATTN) c.shm += (size_t)nw * 16 * KTW * 32 * sizeof(half_t);
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
    if (gemv1_ok(p0)) {
        GemvParams p = p0;
        const Gemv1Cfg c = gemv1_cfg(p);
        p.KTW = c.KTW;
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
            WLX_G1(1, GEMV_IN_XATTN, GEMV_OUT_RESID, 512);
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
                                                            const int* __restrict__ done) {
    __shared__ float prob[WLX_T_TEXT];
    __shared__ int crow[WLX_T_TEXT];
    const int lane = threadIdx.x;
    const int r = blockIdx.x, h = blockIdx.y;
    int dn = 0;
    if (done) dn = *done;
    const int len = pos[r] + 1;
    const short* ar = anc + (long)ancrow[r] * WLX_T_TEXT;
    f16x8 qv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = ld_f16x8(q + (long)r * ldq + h * WLX_HEAD_DIM + i * 8);
    if (dn) return;

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
}

void launch_dec_self_attn(const half_t* q, long ldq, const half_t* Kc, const half_t* Vc, long crs, int d, int H,
                          const RowTables& rt, int rows, half_t* out, long ldo, const int* done, hipStream_t s) {
    if (g_decode_v1)
        hipLaunchKernelGGL(dec_self_attn_kernel, dim3(rows, H), dim3(64), 0, s, q, ldq, Kc, Vc, crs, d, rt.pos,
                           rt.ancrow, rt.anc, out, ldo, done);
    else
        hipLaunchKernelGGL(dec_self_attn2_kernel, dim3(rows, H), dim3(64), 0, s, q, ldq, Kc, Vc, crs, d, rt.pos,
                           rt.ancrow, rt.anc, out, ldo, done);
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
                                                             const int* __restrict__ done) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int sp = blockIdx.x, h = blockIdx.y, grp = blockIdx.z;
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
}

void launch_dec_cross_attn(const half_t* q, long ldq, const half_t* Kx, long ldk, long isk, const half_t* Vtx,
                           long ldvt, long isv, int H, int R, int groups, int rows, const int* group_item,
                           float* part_o, float* part_ml, const int* done, hipStream_t s) {
    if (g_decode_v1)
        hipLaunchKernelGGL(dec_cross_attn_kernel, dim3(WLX_XSPLIT, H, groups), dim3(64), 0, s, q, ldq, Kx, ldk, isk, Vtx,
                           ldvt, isv, H, R, rows, group_item, part_o, part_ml, done);
    else
        hipLaunchKernelGGL(dec_cross_attn2_kernel, dim3(WLX_XSPLIT, H, groups), dim3(64), 0, s, q, ldq, Kx, ldk, isk, Vtx,
                           ldvt, isv, H, R, rows, group_item, part_o, part_ml, done);
}

}  // namespace wlx
