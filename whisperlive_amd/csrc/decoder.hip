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

template <int MT, int NTB>
static void gemv_dispatch_in(const GemvParams& p, dim3 grid, dim3 block, size_t shm, hipStream_t s) {
    switch (p.in_mode) {
        case GEMV_IN_LN: hipLaunchKernelGGL((dec_gemv_kernel<MT, NTB, GEMV_IN_LN>), grid, block, shm, s, p); break;
        case GEMV_IN_F16: hipLaunchKernelGGL((dec_gemv_kernel<MT, NTB, GEMV_IN_F16>), grid, block, shm, s, p); break;
        default: hipLaunchKernelGGL((dec_gemv_kernel<MT, NTB, GEMV_IN_XATTN>), grid, block, shm, s, p); break;
    }
}

void launch_dec_gemv(const GemvParams& p, hipStream_t s) {
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

void launch_dec_self_attn(const half_t* q, long ldq, const half_t* Kc, const half_t* Vc, long crs, int d, int H,
                          const RowTables& rt, int rows, half_t* out, long ldo, const int* done, hipStream_t s) {
    hipLaunchKernelGGL(dec_self_attn_kernel, dim3(rows, H), dim3(64), 0, s, q, ldq, Kc, Vc, crs, d, rt.pos,
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

void launch_dec_cross_attn(const half_t* q, long ldq, const half_t* Kx, long ldk, long isk, const half_t* Vtx,
                           long ldvt, long isv, int H, int R, int groups, int rows, const int* group_item,
                           float* part_o, float* part_ml, const int* done, hipStream_t s) {
    hipLaunchKernelGGL(dec_cross_attn_kernel, dim3(WLX_XSPLIT, H, groups), dim3(64), 0, s, q, ldq, Kx, ldk, isk, Vtx,
                       ldvt, isv, H, R, rows, group_item, part_o, part_ml, done);
}

}  // namespace wlx
