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
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstdio>

namespace wlx {

#ifndef WLX_RESID_BATCHED
#define WLX_RESID_BATCHED 1      // -DWLX_RESID_BATCHED=0 (_lib.build_variant): tile-by-tile residual update (A/B)
#endif
#ifndef WLX_RESID_LN
#define WLX_RESID_LN 4           // residual batches of the large-M form: n-tiles x m-tiles requested together (0 = tile by tile)
#define WLX_RESID_LM 1
#endif

// ---------------- epilogue shared by both GEMM forms: lane owns columns n..n+3 of row m (WNT x WMT accumulator tiles of
// the wave whose first n-tile is nt0 and first row m0)
template <int WNT, int WMT, int MODE>
__device__ __forceinline__ void gemm_epilogue_m(const GemmParams& p, f32x4 (&acc)[WNT][WMT], int nt0, int m0, int z, int c, int g) {
    if constexpr (MODE == GEMM_RESID_F32) {
        // The residual update reads and writes X through one pointer: written tile by tile (load, add, store) every load had to stay behind
        // the previous tile's store — hipcc cannot tell the rows apart (runtime ldx) — so a wave's WNT x WMT tiles were as many DEPENDENT
        // memory round trips. Here the WMT row tiles of an n-tile are requested together (rows past M clamped: no branch around a load), then
        // updated and stored. Measured on one box (profiles/r4resid_residual_epilogue_ab.txt): one window 1.636 -> 1.618 ms per encoder. On the
        // large-M form (WMT = 8, every CU in its epilogue at the same time, 110 MB read-modify-write per launch) WHICH tiles go together
        // matters: the 8 row tiles of an n-tile (128 rows x 64 B) are SLOWER than tile by tile (12 windows 7.54 vs 7.28 ms), the 4 n-tiles of a
        // row tile (16 rows x 256 contiguous bytes) faster (7.12 vs 7.24 ms, large-v3 x 8 25.1 vs 25.5 ms): DRAM page locality.
        // (the large-M form: WLX_RESID_LN x WLX_RESID_LM tiles per batch, -D A/B; 0 = tile by tile)
        constexpr int BN0 = (WMT < 8) ? 1 : WLX_RESID_LN, BM0 = (WMT < 8) ? WMT : WLX_RESID_LM;
        constexpr int BN = BN0 > 0 ? BN0 : 1, BM = BM0 > 0 ? BM0 : 1;
        if constexpr (WLX_RESID_BATCHED && BN0 > 0 && BM0 > 0) {
            static_assert(WNT % BN == 0 && WMT % BM == 0, "batch shape must divide the wave tile");
            float* xb = p.X + (long)z * p.strideX;
            float4 bvs[WNT];
#pragma unroll
            for (int ni = 0; ni < WNT; ++ni) {
                int n = (nt0 + ni) * 16 + g * 4;
                if (n >= p.N) n = 0;                 // (clamped: unused)
                bvs[ni] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int ni0 = 0; ni0 < WNT; ni0 += BN)
#pragma unroll
                for (int mi0 = 0; mi0 < WMT; mi0 += BM) {
                    float4 r[BN][BM];
#pragma unroll
                    for (int a = 0; a < BN; ++a)
#pragma unroll
                        for (int b = 0; b < BM; ++b) {
                            int n = (nt0 + ni0 + a) * 16 + g * 4;
                            if (n >= p.N) n = 0;
                            int m = m0 + (mi0 + b) * 16 + c;
                            if (m >= p.M) m = p.M - 1;
                            r[a][b] = *reinterpret_cast<const float4*>(xb + (long)m * p.ldx + n);
                        }
                    asm volatile("" ::: "memory");      // (keeps the stores below the LAST load of the batch)
#pragma unroll
                    for (int a = 0; a < BN; ++a)
#pragma unroll
                        for (int b = 0; b < BM; ++b) {
                            const int ni = ni0 + a, mi = mi0 + b;
                            const int n = (nt0 + ni) * 16 + g * 4, m = m0 + mi * 16 + c;
                            if (n >= p.N || m >= p.M) continue;
                            const float4 bv = bvs[ni];
                            const float4 o = make_float4(r[a][b].x + (acc[ni][mi][0] + bv.x), r[a][b].y + (acc[ni][mi][1] + bv.y),
                                                         r[a][b].z + (acc[ni][mi][2] + bv.z), r[a][b].w + (acc[ni][mi][3] + bv.w));
                            *reinterpret_cast<float4*>(xb + (long)m * p.ldx + n) = o;
                        }
                }
            return;
        }
    }
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
            switch (MODE) {           // (compile-time: the caller dispatches on p.mode once, not once per accumulator tile)
                case GEMM_STORE_F16:
                case GEMM_GELU_F16: {
                    if (MODE == GEMM_GELU_F16) {
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
template <int WNT, int WMT>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[WNT][WMT], int nt0, int m0, int z, int c, int g) {
    switch (p.mode) {
        case GEMM_STORE_F16: gemm_epilogue_m<WNT, WMT, GEMM_STORE_F16>(p, acc, nt0, m0, z, c, g); break;
        case GEMM_GELU_F16: gemm_epilogue_m<WNT, WMT, GEMM_GELU_F16>(p, acc, nt0, m0, z, c, g); break;
        case GEMM_GELU_POS_F32: gemm_epilogue_m<WNT, WMT, GEMM_GELU_POS_F32>(p, acc, nt0, m0, z, c, g); break;
        case GEMM_RESID_F32: gemm_epilogue_m<WNT, WMT, GEMM_RESID_F32>(p, acc, nt0, m0, z, c, g); break;
        case GEMM_QKV: gemm_epilogue_m<WNT, WMT, GEMM_QKV>(p, acc, nt0, m0, z, c, g); break;
        case GEMM_CROSS_KV: gemm_epilogue_m<WNT, WMT, GEMM_CROSS_KV>(p, acc, nt0, m0, z, c, g); break;
        default: break;
    }
}

// ---------------- LDS-transposed epilogue of the second form (round 2; the phase-removal probe of round 1,
// profiles/r03m, priced the direct epilogue above at 35 % of the MLP-up launch and 27 % of the QKV launch: a wave-level
// store of it covers 16 rows x 32 bytes, V^T and the tile-packed cross-V go out as 2-byte pieces). Here the workgroup's
// 32 WNT x 32 WMT output tile goes through LDS once (the ring is free by then) as fp16 T[m][n] and leaves in the shape the
// destination wants: row-major destinations (C, q, K rows, tile-packed cross-K) as 16-byte pieces, a wave covering 8 full
// 128-byte rows; column-major ones (V^T, tile-packed cross-V) as 8-byte pieces of 4 consecutive rows.
// Values are bit-identical to gemm_epilogue's. The fp32 modes keep the direct form (a lane already owns 16 bytes).
// (NWN x NWM = waves of the workgroup along n / m: 2 x 2 for the second form, 4 x 2 for the third)
template <int WNT, int WMT, int NWN = 2, int NWM = 2>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& p, f32x4 (&acc)[WNT][WMT], half_t* T, int ntb, int mb, int wn, int wm,
                                                  int z, int c, int g, int tid) {
    constexpr int TN = 16 * NWN * WNT, TM = 16 * NWM * WMT, PT = TN + 8, NTHR = 64 * NWN * NWM;   // tile columns / rows, LDS pitch in halfs (16-byte aligned rows)
    const int n_wg = ntb * 16;                                        // first column of the workgroup (multiple of 32 WNT)
    // what the workgroup's columns are (uniform: d_model is a multiple of the tile width)
    int part = 0;                                                     // QKV: 0 q, 1 k, 2 v;  CROSS_KV: 0 k, 1 v
    int layer = 0, col0 = n_wg;                                       // column inside its part
    if (p.mode == GEMM_QKV) { part = n_wg / p.d; col0 = n_wg - part * p.d; }
    else if (p.mode == GEMM_CROSS_KV) { layer = n_wg / (2 * p.d); const int nn = n_wg - layer * 2 * p.d; part = nn / p.d; col0 = nn - part * p.d; }
    const float scale = (p.mode == GEMM_QKV && part == 0) ? p.qscale : 1.0f;
#pragma unroll
    for (int ni = 0; ni < WNT; ++ni) {
        const int nl = (wn * WNT + ni) * 16 + g * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && n_wg + nl < p.N) bv = *reinterpret_cast<const float4*>(p.bias + n_wg + nl);
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi) {
            const int ml = (wm * WMT + mi) * 16 + c;
            float v0 = acc[ni][mi][0] + bv.x, v1 = acc[ni][mi][1] + bv.y, v2 = acc[ni][mi][2] + bv.z, v3 = acc[ni][mi][3] + bv.w;
            if (p.mode == GEMM_GELU_F16) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
            const f16x4 o = {(half_t)(v0 * scale), (half_t)(v1 * scale), (half_t)(v2 * scale), (half_t)(v3 * scale)};
            *reinterpret_cast<f16x4*>(T + ml * PT + nl) = o;
        }
    }
    __syncthreads();
    const bool colmajor = (p.mode == GEMM_QKV && part == 2) || (p.mode == GEMM_CROSS_KV && part == 1);
    if (!colmajor) {
        // 16-byte pieces: (row, 8-column segment). Row-major destinations: segment fastest (a wave = 8 full rows);
        // tile-packed cross-K: row fastest (16 consecutive keys of a segment are 256 contiguous bytes of the image)
        constexpr int SEG = TN / 8, UNITS = TM * SEG;
        for (int u = tid; u < UNITS; u += NTHR) {
            int row, seg;
            if (p.mode == GEMM_CROSS_KV) { seg = u / TM; row = u - seg * TM; } else { row = u / SEG; seg = u - row * SEG; }
            const int m = mb + row, n = n_wg + seg * 8;
            if (m >= p.M || n >= p.N) continue;
            const f16x8 v = *reinterpret_cast<const f16x8*>(T + row * PT + seg * 8);
            half_t* dst;
            if (p.mode == GEMM_QKV) {
                if (part == 0) dst = p.C + (long)m * p.ldc + n;
                else {
                    const int item = m / p.rows_per_item, t = m - item * p.rows_per_item;
                    dst = p.Kout + (long)item * p.kv_item_stride_k + (long)t * p.ldk + (col0 + seg * 8);
                }
            } else if (p.mode == GEMM_CROSS_KV) {
                const int item = m / p.rows_per_item, t = m - item * p.rows_per_item;
                const int da = col0 + seg * 8, hd = da >> 6, dd = da & 63;
                const int tile = t >> 5, k32 = t & 31;
                const long tbase = ((long)hd * (WLX_T_AUDIO_PAD / 32) + tile) * 2048;
                const int s2 = k32 >> 4, cc = k32 & 15, kt2 = dd >> 5, gg = (dd & 31) >> 3;
                dst = p.Kout + (long)layer * p.kv_layer_stride_k + (long)item * p.kv_item_stride_k + tbase + ((s2 * 2 + kt2) * 64 + gg * 16 + cc) * 8;
            } else dst = p.C + (long)z * p.strideC + (long)m * p.ldc + n;
            *reinterpret_cast<f16x8*>(dst) = v;
        }
    } else {
        // 8-byte pieces: (column, 4 consecutive rows); rows_per_item is a multiple of 4, so a piece never straddles two items
        constexpr int MG = TM / 4, UNITS = TN * MG;
        for (int u = tid; u < UNITS; u += NTHR) {
            int col, mg;
            if (p.mode == GEMM_CROSS_KV) { mg = u / TN; col = u - mg * TN; } else { col = u / MG; mg = u - col * MG; }
            const int m0 = mb + mg * 4, n = n_wg + col;
            if (m0 >= p.M || n >= p.N) continue;
            const half_t* tp = T + (mg * 4) * PT + col;
            const f16x4 v = {tp[0], tp[PT], tp[2 * PT], tp[3 * PT]};
            const int item = m0 / p.rows_per_item, t = m0 - item * p.rows_per_item;
            half_t* dst;
            if (p.mode == GEMM_QKV) dst = p.Vt + (long)item * p.kv_item_stride_v + (long)(col0 + col) * p.ldvt + t;
            else {
                const int da = col0 + col, hd = da >> 6, dd = da & 63;
                const int tile = t >> 5, k32 = t & 31;
                const long tbase = ((long)hd * (WLX_T_AUDIO_PAD / 32) + tile) * 2048;
                const int dt = dd >> 4, c0 = dd & 15, gg = (k32 & 15) >> 2, ee = (k32 >> 4) << 2;
                dst = p.Vt + (long)layer * p.kv_layer_stride_v + (long)item * p.kv_item_stride_v + tbase + (dt * 64 + gg * 16 + c0) * 8 + ee;
            }
            if (m0 + 3 < p.M) *reinterpret_cast<f16x4*>(dst) = v;
            else {                                                  // ragged last rows (M not a multiple of 4): element by element
                for (int j = 0; j < 4 && m0 + j < p.M; ++j) {
                    if (p.mode == GEMM_QKV) dst[j] = v[j];
                    else {
                        const int tj = t + j, k32j = tj & 31;
                        const int da = col0 + col, hd = da >> 6, dd = da & 63;
                        const long tb = ((long)hd * (WLX_T_AUDIO_PAD / 32) + (tj >> 5)) * 2048;
                        p.Vt[(long)layer * p.kv_layer_stride_v + (long)item * p.kv_item_stride_v + tb +
                             ((dd >> 4) * 64 + ((k32j & 15) >> 2) * 16 + (dd & 15)) * 8 + (k32j & 3) + ((k32j >> 4) << 2)] = v[j];
                    }
                }
            }
        }
    }
}

// (The first form — register-staged double buffer, one stage ahead — was the A/B reference of round 2 and is no longer in the library
// since round 5; the second form below is bit-identical to it, DESIGN.md "Encoder GEMM".)

// ------------------------------------------------------------------------------------------------------------------
// Second form (round 2): the same tiles, fragments, MFMA order and epilogue — bit-identical results — with the stage loop
// rebuilt around what the ISA of the first form showed: its global loads for stage s+1 are issued one stage (~0.25 us of
// MFMAs) before `s_waitcnt vmcnt(0)`, so every one of the 12-96 stages of a workgroup still pays most of an L2/HBM round
// trip, and at ~1.1 workgroups per CU nothing else hides it. Here both operands go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: the per-lane SOURCE address is the fragment shape, the LDS image is lane-linear = fragment
// order, so the ds_read_b128 of a fragment stays conflict-free) into a ring of DEPTH stages; DEPTH-1 stages are in flight
// ahead of the MFMAs, each stage is released by a COUNTED vmcnt (this wave's pieces of the oldest stage) + one raw
// s_barrier (everyone's pieces; also: everyone is done reading the buffer that is refilled next). `__syncthreads()` must
// not be used here: with LDS-DMA in flight hipcc lowers it to vmcnt(0) + s_barrier and the ring drains every stage.
typedef __attribute__((address_space(3))) void wlx_lds_void;

// KS = k-tiles (of 32) per stage. (KS = 4 — 128-deep stages, a 120 KiB ring, one workgroup per CU — was measured in round 5 for the
// N = d_model projections of one window, 192 workgroups whose K = 4 d_model loop takes 24 us against 10 us at K = d_model: 17.44 vs
// 17.49 us per launch, profiles/r5f_*. Halving the number of stages and barriers changes nothing: the loop's time goes with the BYTES a
// workgroup pulls through its CU — ~0.25 us per k-tile of a 64 x 96 tile = ~40 GB/s per CU — not with its stage count.)
template <int WNT, int WMT, int DEPTH, int KS = 2>
__global__ __launch_bounds__(256) void gemm2_kernel(GemmParams p) {
    constexpr int FA = 2 * WNT * KS, FB = 2 * WMT * KS;
    constexpr int CA = FA / 4, CB = FB / 4;     // fragments (1 KiB LDS-DMA pieces) each wave requests per stage
    constexpr int NL = CA + CB;                 // = this wave's vmcnt events per stage
    static_assert(DEPTH >= 2 && NL * (DEPTH - 2) <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) f16x8 ring[];        // [DEPTH][FA + FB][64 lanes] x 16 B — the ONLY LDS object
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int wn = wave & 1, wm = wave >> 1;
    const int NT_total = (p.N + 15) >> 4;
    // XCD-aware tile map (launcher: p.xcd_a x p.xcd_b = 8, 0 = off). Workgroups are dealt round-robin over the 8 XCDs
    // (linear id % 8) and every XCD has its own 4 MiB L2: with the plain (x fastest) map each XCD's workgroups touch
    // EVERY weight panel and EVERY activation panel — 7-14 MB per encoder GEMM through a 4 MiB L2, so most stage fills
    // are L2 misses (measured: TCC miss bytes 4-8x the unique operand bytes) and the LDS-DMA ring waits on HBM latency.
    // Here XCD j owns the sub-rectangle (m-part j / b, n-part j % b) of the tile grid: its working set is 1/a of the
    // activations plus 1/b of the weights.
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_a > 0) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int lin = bx + gx * by;
        const int xcd = lin & 7, k = lin >> 3;
        const int a = p.xcd_a, b = p.xcd_b;                 // a | gy, b | gx, (gx * gy) % 8 == 0 (launcher)
        const int sy = gy / a, sx = gx / b;                  // tiles of one XCD: sy x sx, sy * sx == gx * gy / 8
        const int ja = xcd / b, jb = xcd - ja * b;
        const int ly = k % sy, lx = k / sy;
        by = ja * sy + ly;
        bx = jb * sx + lx;
    }
    const int ntb = bx * (2 * WNT);
    const int mb = by * (2 * WMT * 16);
    const int nt0 = ntb + wn * WNT;
    const int m0 = mb + wm * (WMT * 16);
    const int z = blockIdx.z;
    const half_t* A = p.A + (long)z * p.strideA;
    const int KT = p.KT;

    const half_t* asrc[CA];
    const half_t* bsrc[CB];
#pragma unroll
    for (int j = 0; j < CA; ++j) {
        const int f = wave + 4 * j, ni = f / KS, kk = f % KS;
        int nt = ntb + ni;
        if (nt >= NT_total) nt = NT_total - 1;
        asrc[j] = p.Wp + ((long)nt * KT + kk) * 512 + lane * 8;
    }
    // Activation pieces (round 6): FULL 128-byte row segments — a piece = 8 rows x 128 B = both k-tiles of the stage — into the swizzled image the
    // large-M form uses (LDS row p = (r % 8) * 2 + r / 8 of a 16-row tile holds row r, 16-byte slot s holds segment s ^ (r & 7); the permutation sits on
    // the per-lane SOURCE address, the LDS side of an LDS-DMA is lane-linear). The fragment-shaped pieces this form had (16 rows x 64 B, one k-tile)
    // touch 16 cache lines per wave-level request for the same KiB. Same fragments into the same MFMAs: identical results.
    static_assert(KS == 2, "a stage = two k-tiles = one 128-byte segment per activation row");
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int f = wave + 4 * j, mi = f >> 1, ph = f & 1;
        const int r = ((lane >> 3) & 1) * 8 + ph * 4 + (lane >> 4);           // row of the 16-row tile this lane fetches
        int row = mb + mi * 16 + r;
        if (row >= p.M) row = p.M - 1;
        bsrc[j] = A + (long)row * p.lda + (((lane & 7) ^ (r & 7)) << 3);
    }
    const int boff0 = (((c & 7) << 1) + (c >> 3)) * 8 + ((0 + g) ^ (c & 7));    // f16x8 units inside a 16-row tile image: k-tile 0 / 1 of the stage
    const int boff1 = (((c & 7) << 1) + (c >> 3)) * 8 + ((4 + g) ^ (c & 7));
    auto issue = [&](int st, int buf) {         // this wave's pieces of stage `st` -> ring buffer `buf`
        f16x8* dst = ring + (long)buf * (FA + FB) * 64;
#pragma unroll
        for (int j = 0; j < CA; ++j)
            __builtin_amdgcn_global_load_lds((const void*)(asrc[j] + (long)st * (KS * 512)), (wlx_lds_void*)(dst + (wave + 4 * j) * 64), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < CB; ++j)
            __builtin_amdgcn_global_load_lds((const void*)(bsrc[j] + (long)st * (KS * 32)), (wlx_lds_void*)(dst + (FA + wave + 4 * j) * 64), 16, 0, 0);
    };

    f32x4 acc[WNT][WMT];
#pragma unroll
    for (int ni = 0; ni < WNT; ++ni)
#pragma unroll
        for (int mi = 0; mi < WMT; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int S = KT / KS;                      // the launcher only picks this form for KT % KS == 0
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i)
        if (i < S) issue(i, i);
    int rbuf = 0, wbuf = DEPTH - 1;             // buffer read this trip / buffer refilled this trip (the one read last trip)
#pragma unroll 1
    for (int st = 0; st < S; ++st) {
        // stages st .. min(st + DEPTH - 2, S - 1) are outstanding; stage st must have landed
        if (st + DEPTH - 2 < S) __builtin_amdgcn_s_waitcnt(((NL * (DEPTH - 2)) & 15) | (((NL * (DEPTH - 2)) >> 4) << 14) | 0x0F70);
        else __builtin_amdgcn_s_waitcnt(0x0F70);                        // tail: fewer stages in flight, drain
        __builtin_amdgcn_s_barrier();
        if (st + DEPTH - 1 < S) issue(st + DEPTH - 1, wbuf);
        const f16x8* src = ring + (long)rbuf * (FA + FB) * 64 + lane;
        const f16x8* bsw = ring + (long)rbuf * (FA + FB) * 64 + FA * 64;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            f16x8 wf[WNT], af[WMT];
#pragma unroll
            for (int ni = 0; ni < WNT; ++ni) wf[ni] = src[((wn * WNT + ni) * KS + kk) * 64];
#pragma unroll
            for (int mi = 0; mi < WMT; ++mi) af[mi] = bsw[(wm * WMT + mi) * 128 + (kk ? boff1 : boff0)];
#pragma unroll
            for (int ni = 0; ni < WNT; ++ni)
#pragma unroll
                for (int mi = 0; mi < WMT; ++mi) acc[ni][mi] = mfma16(wf[ni], af[mi], acc[ni][mi]);
        }
        rbuf = (rbuf + 1 == DEPTH) ? 0 : rbuf + 1;
        wbuf = (wbuf + 1 == DEPTH) ? 0 : wbuf + 1;
    }
    // (no LDS-DMA is in flight any more: the tail of the loop drained it, so __syncthreads() is safe from here on)
    const bool f16_out = p.mode == GEMM_STORE_F16 || p.mode == GEMM_GELU_F16 || p.mode == GEMM_QKV || p.mode == GEMM_CROSS_KV;
    if (p.epi_lds && f16_out && (p.rows_per_item & 3) == 0 && (p.mode == GEMM_STORE_F16 || p.mode == GEMM_GELU_F16 || p.d % (32 * WNT) == 0)) {
        __syncthreads();                                    // everyone is done reading the last stage: the ring becomes the tile
        gemm_epilogue_lds<WNT, WMT>(p, acc, reinterpret_cast<half_t*>(ring), ntb, mb, wn, wm, z, c, g, tid);
    } else
        gemm_epilogue<WNT, WMT>(p, acc, nt0, m0, z, c, g);
}

// ------------------------------------------------------------------------------------------------------------------
// Third form (round 4): the LARGE-M GEMM. The second form's 64 x 96 tile was tuned for M = 1500 (one window: ~1.1
// workgroup rounds, latency-bound); a batched encoder (M = 1500 B, B = 4..12) is throughput-bound and ran the same tile at
// 15 % of the MFMA peak. Here: 256 (m) x 256 (n) x 64 (k) workgroup tiles, 8 waves as 2 (m) x 4 (n), 128 x 64 outputs per wave
// (128 accumulator registers), and the K loop as FOUR PHASES per 64-deep K tile (the guide's 8-phase schedule = two K tiles
// per trip): each phase reads one register subtile from LDS, requests one 16 KiB half-tile of a LATER K tile by LDS-DMA,
// and runs the 16 MFMAs of one quadrant of the wave's outputs. The two wave groups (waves 0-3 / 4-7: one of each per SIMD) run
// half a phase apart (one extra s_barrier), so on every SIMD one wave is in its MFMA burst while its partner issues LDS reads and
// DMA requests. LDS = 2 K-tile buffers x {X0, W0, W1, X1} half-tiles of 16 KiB = 128 KiB:
//   X half h = for each wave row wm the 64 activation rows wm*128 + h*64 .. +63   (read by the phase that starts quadrant row h)
//   W half h = for each wave column wn the 32 weight rows wn*64 + h*32 .. +31
// Phase order per K tile (q = (quadrant row, quadrant column)): P0 reads X0 + W0 -> q(0,0); P1 reads W1 -> q(0,1); P2 reads X1 ->
// q(1,1); P3 -> q(1,0) from registers. So every half-tile is read in exactly ONE phase and its slot is re-staged two phases later:
//   P0 requests W1(t+1), P1 X1(t+1), P2 X0(t+2), P3 W0(t+2); the only wait is `vmcnt(4)` in P3 (this wave's two newest
//   half-tiles stay in flight): everything of K tile t+1 has landed before the barrier that precedes its first read.
// RAW: wait in phase p (both groups, before a barrier), read in phase p+1. WAR: re-stage >= 2 phases after the last read (the
// staggered group's reads are complete — lgkmcnt(0) — one barrier before the re-staging request is issued).
// Weights stay in their packed fragment order (1 KiB per DMA piece, lane-linear, conflict-free ds_read_b128). Activations are
// fetched as FULL 128-byte row segments (a DMA piece = 8 rows x 128 B; the second form's fragment-shaped pieces, 16 rows x 64 B,
// cost the texture path twice the cycles and this tile runs it at ~50 % of its rate) into a swizzled image: LDS row
// p = (r % 8) * 2 + r / 8 of a 16-row tile holds global row r, 16-byte slot s holds segment s ^ (r & 7) — the permutation is
// applied to the per-lane SOURCE address (the LDS side of an LDS-DMA is lane-linear) and again on the read; ds_read_b128's four
// 16-lane groups ({0-3,12-15,20-27}, ...) then touch 16 distinct slots of the 256-byte bank row.
// K tiles past the end are requested again from the last K tile (clamped): the slots they land in are never read, the request
// count per phase — and with it every vmcnt — stays uniform; the cost is two K tiles of L2 hits per workgroup.
#define G3_XH 16384
// PERSISTENT form (round 4, second step): one workgroup per CU walks its share of the tile list. With one workgroup per CU (128 KiB of
// LDS, 256 registers) nothing overlapped a tile's prologue (first K tile: an HBM round trip) and epilogue with another tile's K loop:
// per-launch traces at 12 windows showed ~40 us per tile for the K = 768 GEMMs whose 12 K tiles need ~17 us. Now the NEXT tile's
// prologue requests are issued before the epilogue of the finished one wherever the epilogue does not need the ring: fp32 modes store
// from registers; the fp16 row-major modes (MLP up-projection) go through a WAVE-PRIVATE 4.5 KiB staging area in the dead W1 / X1 slots
// of buffer 1 (32 rows x 64 columns at a time -> 16-byte pieces, a wave store = 8 full 128-byte rows; no workgroup barrier) — also q,
// K rows and the tile-packed cross K of the scattering modes, whose pieces are 16 bytes of one row as well. The column-major outputs
// (V^T for the encoder attention, the tile-packed cross V) need no transposition at all: for a tile whose 256 columns are V columns
// (uniform per tile: d_model % 256 == 0) the MFMAs run with their operands SWAPPED — the A and B fragments of 16x16x32 have the same
// register layout (row c, 8 k at g*8), so mfma(x, w) instead of mfma(w, x) yields the transposed accumulator tile: a lane then holds
// 4 consecutive ROWS (time steps) of one column, i.e. one 8-byte piece of V^T / of the packed V image, stored straight from registers.
// (Same products, same k order: the values are those of the untransposed tile.) Probes of the first persistent version (profiles/r4h_*:
// K loop only / epilogue only) had the workgroup-wide LDS-transposed epilogue at 15 us per QKV tile against a 20 us K loop.
// The epilogue's stores share the vmcnt queue with the LDS-DMA requests, so a tile starts with vmcnt(0) instead of a counted wait.
__global__ __launch_bounds__(512) void gemm3_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char g3lds[];           // [2][X0 | W0 | W1 | X1] x 16 KiB (+ 4 KiB) — the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // ---- tile list: n fastest; each XCD (workgroup id % 8) owns a contiguous run of it (its L2 holds that run's activation panels and —
    // N / 256 <= 12 panels for the layer GEMMs — all the weight panels); the XCD's workgroups walk the run together
    const int gx = p.g3_gx, ntiles = p.g3_tiles;
    const int xcd = (int)blockIdx.x & 7, nper = (int)gridDim.x >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int run0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, runlen = q8 + (xcd < r8 ? 1 : 0);
    int kpos = (int)blockIdx.x >> 3;                                        // position in the XCD's run
    if (kpos >= runlen) return;
    const int KT = p.KT, NK = KT >> 1;
    // LDS-DMA requests go through buffer descriptors (buffer_load_dwordx4 ... lds): the per-lane part of an address is ONE 32-bit
    // register per distinct lane pattern (4 for the activations, 1 for the weights), everything else — tile, K position — is a scalar
    // offset. (As 64-bit global addresses the eight source pointers and their per-request adds pushed the kernel into scratch.)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, 0x7fffffff, 0x00020000);
    int xvo[2][2];                              // per-lane byte offsets of the four activation pieces this wave requests per K tile
    int wso[2][2];                              // scalar byte offsets of its four weight pieces (K tile 0)
    const int wvo = lane * 16;
    int n_blk = 0, m_blk = 0;
    // ---- LDS-DMA sources of a tile. A half-tile = 16 pieces of 1 KiB; wave w requests pieces w and w + 8.
    //   X half h, piece j: 16-row tile mi = j >> 1 of the half (wave row mi >> 2, tile mi & 3), LDS rows (j & 1) * 8 .. + 7
    //   W half h, piece j: wave column j >> 2, n-tile (j >> 1) & 1 of its half, k-tile j & 1 of the K tile
    auto set_tile = [&](int lin) {
        const int by = lin / gx, bx = lin - by * gx;
        n_blk = bx * 256; m_blk = by * 256;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = wave + 8 * q;
                const int mi = j >> 1, ph = j & 1;
                const int r = ((lane >> 3) & 1) * 8 + ph * 4 + (lane >> 4);  // global row of the 16-row tile this lane fetches
                int row = m_blk + (mi >> 2) * 128 + h * 64 + (mi & 3) * 16 + r;
                if (row >= p.M) row = p.M - 1;
                xvo[h][q] = (int)(((long)row * p.lda + (((lane & 7) ^ (r & 7)) << 3)) * 2);      // (launcher: M * lda * 2 < 2^31)
                const int ntile = (n_blk >> 4) + (j >> 2) * 4 + h * 2 + ((j >> 1) & 1);
                wso[h][q] = __builtin_amdgcn_readfirstlane((int)(((long)ntile * KT + (j & 1)) * 1024));   // (launcher: N * K * 2 < 2^31)
            }
    };
    // slot of half-tile `part` (0 X0, 1 W0, 2 W1, 3 X1) in K-tile buffer b; piece j at + j KiB
    auto dma_x = [&](int b, int part, int h, int kt64) {
        const int kk = kt64 < NK ? kt64 : NK - 1;
        char* dst = g3lds + b * 65536 + part * G3_XH;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (wlx_lds_void*)(dst + (wave + 8 * q) * 1024), 16, xvo[h][q], kk * 128, 0, 0);
    };
    auto dma_w = [&](int b, int part, int h, int kt64) {
        const int kk = kt64 < NK ? kt64 : NK - 1;
        char* dst = g3lds + b * 65536 + part * G3_XH;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (wlx_lds_void*)(dst + (wave + 8 * q) * 1024), 16, wvo, wso[h][q] + kk * 2048, 0, 0);
    };
    auto prologue = [&]() {                     // K tile 0 complete, X0 / W0 of K tile 1
        dma_x(0, 0, 0, 0); dma_w(0, 1, 0, 0); dma_w(0, 2, 1, 0); dma_x(0, 3, 1, 0);
        dma_x(1, 0, 0, 1); dma_w(1, 1, 0, 1);
    };
    // ---- LDS read offsets of this lane
    const int xrow = (((c & 7) << 1) + (c >> 3)) * 128 + wm * 8192;          // LDS row p(c) of the wave row's first tile
    const int xo0 = xrow + (((0 + g) ^ (c & 7)) << 4), xo1 = xrow + (((4 + g) ^ (c & 7)) << 4);   // k-tile 0 / 1 of the K tile
    const int wo = wn * 4096 + lane * 16;
    f16x8 xr[4][2], w0r[2][2], w1r[2][2];       // register subtiles: X half (4 m-tiles x 2 k-tiles), W halves (2 n-tiles x 2 k-tiles)
    auto read_x = [&](int b, int part) {
        const char* base = g3lds + b * 65536 + part * G3_XH;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            xr[mt][0] = *reinterpret_cast<const f16x8*>(base + mt * 2048 + xo0);
            xr[mt][1] = *reinterpret_cast<const f16x8*>(base + mt * 2048 + xo1);
        }
    };
    auto read_w = [&](int b, int part, f16x8 (&wr)[2][2]) {
        const char* base = g3lds + b * 65536 + part * G3_XH + wo;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) wr[nt][kt] = *reinterpret_cast<const f16x8*>(base + (nt * 2 + kt) * 1024);
    };
    const bool f16_out = p.mode == GEMM_STORE_F16 || p.mode == GEMM_GELU_F16 || p.mode == GEMM_QKV || p.mode == GEMM_CROSS_KV;
    const bool staged = p.epi_lds && f16_out;   // (launcher: rows_per_item % 4 == 0 and d_model % 256 == 0 for the scattering modes)
    // what the tile's 256 columns are (uniform): QKV part 0 q / 1 k / 2 v; cross K/V: decoder layer, part 0 k / 1 v
    int part = 0, layer = 0, col0 = 0;
    bool tr = false;                            // this tile accumulates transposed (V columns)
    auto classify = [&]() {
        part = 0; layer = 0; col0 = n_blk;
        if (p.mode == GEMM_QKV) { part = n_blk / p.d; col0 = n_blk - part * p.d; }
        else if (p.mode == GEMM_CROSS_KV) { layer = n_blk / (2 * p.d); const int nn = n_blk - layer * 2 * p.d; part = nn / p.d; col0 = nn - part * p.d; }
        tr = staged && ((p.mode == GEMM_QKV && part == 2) || (p.mode == GEMM_CROSS_KV && part == 1));
    };

    set_tile(run0 + kpos);
    prologue();
#pragma unroll 1
    for (;;) {
        classify();
        f32x4 acc[4][8];                        // [n-tile of the wave: quadrant column * 2 + tile][m-tile: quadrant row * 4 + tile]
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto quad = [&](int qn, int qm, f16x8 (&wr)[2][2]) {     // 16 MFMAs: k-tile outermost (dependent pairs 8 apart)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[qn * 2 + nt][qm * 4 + mt] = mfma16(wr[nt][kt], xr[mt][kt], acc[qn * 2 + nt][qm * 4 + mt]);
        };
        auto quad_t = [&](int qn, int qm, f16x8 (&wr)[2][2]) {   // the same tile transposed: D[i = m][j = n], lane = (column c, rows g*4 ..)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[qn * 2 + nt][qm * 4 + mt] = mfma16(xr[mt][kt], wr[nt][kt], acc[qn * 2 + nt][qm * 4 + mt]);
        };
        const bool trq = tr;                    // (uniform; read once per tile) — the K loop exists twice, selected per tile: a branch per
                                                // phase between the two MFMA orders costs registers (both operand orders live) and spilled
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's prologue pieces (and the previous tile's stores)
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs half a phase behind
#define G3_MID()  do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_waitcnt(0xC07F); \
                       __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); } while (0)
#define G3_END()  do { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
        auto kloop = [&](auto TRc) {
        constexpr bool TR = decltype(TRc)::value;
#pragma unroll 1
        for (int t = 0; t < NK; t += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {       // K tile t + u lives in buffer u (NK is even: launcher)
                const int tt = t + u;
                // P0: X0 + W0 -> quadrant (0, 0); request W1(t+1)
                read_w(u, 1, w0r); __builtin_amdgcn_sched_barrier(0); read_x(u, 0);
                dma_w(u ^ 1, 2, 1, tt + 1);
                G3_MID(); if constexpr (TR) quad_t(0, 0, w0r); else quad(0, 0, w0r); G3_END();
                // P1: W1 -> quadrant (0, 1); request X1(t+1)
                read_w(u, 2, w1r);
                dma_x(u ^ 1, 3, 1, tt + 1);
                G3_MID(); if constexpr (TR) quad_t(1, 0, w1r); else quad(1, 0, w1r); G3_END();
                // P2: X1 -> quadrant (1, 1); request X0(t+2)
                read_x(u, 3);
                dma_x(u, 0, 0, tt + 2);
                G3_MID(); if constexpr (TR) quad_t(1, 1, w1r); else quad(1, 1, w1r); G3_END();
                // P3: quadrant (1, 0) from registers; request W0(t+2); everything of K tile t+1 must have landed
                dma_w(u, 1, 0, tt + 2);
                __builtin_amdgcn_s_waitcnt(0x0F74);                         // vmcnt(4)
                G3_MID(); if constexpr (TR) quad_t(0, 1, w0r); else quad(0, 1, w0r); G3_END();
            }
        }
        };
        if (trq) kloop(std::true_type{}); else kloop(std::false_type{});
#undef G3_MID
#undef G3_END
        if (wm == 0) __builtin_amdgcn_s_barrier();  // (pairs with the second group's last phase barrier): every read of the ring is done
        // ---- this tile's outputs; the next tile's first K tiles are requested first (no epilogue needs the ring any more)
        const int nt0 = (n_blk >> 4) + wn * 4, m0 = m_blk + wm * 128;
        const int e_part = part, e_layer = layer, e_col0 = col0 + wn * 64;    // (of the finished tile; e_col0: this wave's first column inside its part)
        kpos += nper;
        const bool more = kpos < runlen;
        if (more) { set_tile(run0 + kpos); prologue(); }                    // (the clamped tail requests of this wave land before these: same wave, in order)
        if (!staged) {
            gemm_epilogue<4, 8>(p, acc, nt0, m0, 0, c, g);
        } else if (trq) {
            // transposed accumulators: lane (c, g) holds rows m = m-tile + g*4 .. +3 of column nt*16 + c: an 8-byte piece of V^T (QKV) or of
            // the tile-packed cross V (4 consecutive keys of one dim). rows_per_item % 4 == 0: a piece never straddles two items.
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int da = e_col0 + ni * 16 + c;                        // column inside V (dim of the model)
                const float bn = p.bias ? p.bias[(nt0 + ni) * 16 + c] : 0.f;
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    const int m = m0 + mi * 16 + g * 4;
                    if (m >= p.M) continue;
                    const f32x4 a = acc[ni][mi];
                    const f16x4 o = {(half_t)(a[0] + bn), (half_t)(a[1] + bn), (half_t)(a[2] + bn), (half_t)(a[3] + bn)};
                    const int item = m / p.rows_per_item, t = m - item * p.rows_per_item;
                    half_t* dst;
                    if (p.mode == GEMM_QKV) dst = p.Vt + (long)item * p.kv_item_stride_v + (long)da * p.ldvt + t;
                    else {
                        const int hd = da >> 6, dd = da & 63, tile = t >> 5, k32 = t & 31;
                        dst = p.Vt + (long)e_layer * p.kv_layer_stride_v + (long)item * p.kv_item_stride_v + ((long)hd * (WLX_T_AUDIO_PAD / 32) + tile) * 2048 +
                              ((dd >> 4) * 64 + ((k32 & 15) >> 2) * 16 + (dd & 15)) * 8 + ((k32 >> 4) << 2);
                    }
                    if (m + 3 < p.M) *reinterpret_cast<f16x4*>(dst) = o;
                    else {                                                  // ragged last rows (M not a multiple of 4): element by element
                        for (int j = 0; j < 4 && m + j < p.M; ++j) {
                            if (p.mode == GEMM_QKV) dst[j] = o[j];
                            else {
                                const int tj = t + j, hd = da >> 6, dd = da & 63, k32j = tj & 31;
                                p.Vt[(long)e_layer * p.kv_layer_stride_v + (long)item * p.kv_item_stride_v + ((long)hd * (WLX_T_AUDIO_PAD / 32) + (tj >> 5)) * 2048 +
                                     ((dd >> 4) * 64 + ((k32j & 15) >> 2) * 16 + (dd & 15)) * 8 + (k32j & 3) + ((k32j >> 4) << 2)] = o[j];
                            }
                        }
                    }
                }
            }
        } else {
            // row-major pieces through a wave-private staging area in buffer 1's W1 / X1 slots (+ the 4 KiB above the ring): neither the
            // tail requests of the finished K loop nor the next tile's prologue touch them. 32 rows x 64 columns at a time; a lane then
            // takes 16 bytes = 8 consecutive columns of one row: fp16 rows of C / q / K, or one lane-slot of the tile-packed cross K.
            half_t* stg = reinterpret_cast<half_t*>(g3lds + 98304 + wave * 4608);   // [32 rows][72] fp16 (64 columns + 16 bytes of pitch padding)
            const float scale = (p.mode == GEMM_QKV && e_part == 0) ? p.qscale : 1.0f;
            float4 bv[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bv[ni] = p.bias ? *reinterpret_cast<const float4*>(p.bias + (nt0 + ni) * 16 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {                                // 2 m-tiles = 32 rows of the wave's 128 at a time
#pragma unroll
                for (int mi2 = 0; mi2 < 2; ++mi2)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const f32x4 a = acc[ni][ch * 2 + mi2];
                        float v0 = a[0] + bv[ni].x, v1 = a[1] + bv[ni].y, v2 = a[2] + bv[ni].z, v3 = a[3] + bv[ni].w;
                        if (p.mode == GEMM_GELU_F16) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
                        const f16x4 o = {(half_t)(v0 * scale), (half_t)(v1 * scale), (half_t)(v2 * scale), (half_t)(v3 * scale)};
                        *reinterpret_cast<f16x4*>(stg + (mi2 * 16 + c) * 72 + ni * 16 + g * 4) = o;
                    }
                __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0): this wave's own writes (LDS serves a wave in order)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + (lane >> 3), seg = lane & 7;
                    const f16x8 v = *reinterpret_cast<const f16x8*>(stg + row * 72 + seg * 8);
                    const int m = m0 + ch * 32 + row;
                    if (m >= p.M) continue;
                    half_t* dst;
                    if (p.mode == GEMM_QKV) {
                        if (e_part == 0) dst = p.C + (long)m * p.ldc + nt0 * 16 + seg * 8;
                        else {
                            const int item = m / p.rows_per_item, t = m - item * p.rows_per_item;
                            dst = p.Kout + (long)item * p.kv_item_stride_k + (long)t * p.ldk + (e_col0 + seg * 8);
                        }
                    } else if (p.mode == GEMM_CROSS_KV) {
                        const int item = m / p.rows_per_item, t = m - item * p.rows_per_item;
                        const int da = e_col0 + seg * 8, hd = da >> 6, dd = da & 63, tile = t >> 5, k32 = t & 31;
                        dst = p.Kout + (long)e_layer * p.kv_layer_stride_k + (long)item * p.kv_item_stride_k + ((long)hd * (WLX_T_AUDIO_PAD / 32) + tile) * 2048 +
                              (((k32 >> 4) * 2 + (dd >> 5)) * 64 + ((dd & 31) >> 3) * 16 + (k32 & 15)) * 8;
                    } else dst = p.C + (long)m * p.ldc + nt0 * 16 + seg * 8;
                    *reinterpret_cast<f16x8*>(dst) = v;
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);                         // the reads are done before the next chunk overwrites the area
            }
        }
        if (!more) break;
    }
}
#define G3_LDS_BYTES (256 * (256 + 8) * 2)      // the fp16 output tile of the LDS-transposed epilogue (135168 B) >= the 128 KiB ring
// WLX_GEMM3 (A/B builds, -DWLX_AB: common.h wlx_ab): 0 = off, 2 = every GEMM with M >= 256 on this form (the full-depth parity tests of
// a single window are run once that way: the form is then held to the single-window path bit for bit)
static bool gemm3_ok(const GemmParams& p, int zbatch) {
    static const int mode = [] { const char* e = wlx_ab("WLX_GEMM3"); return e ? atoi(e) : 1; }();
    if (mode == 0 || zbatch != 1 || (p.N & 255) || (p.KT & 3) || p.KT < 8) return false;
    if ((long)p.M * p.lda * 2 >= (1L << 31) || (long)p.N * p.KT * 64 >= (1L << 31)) return false;   // 32-bit buffer offsets
    // measured (profiles/r4k_encode_shape_times.txt, Whisper-small): 2 windows (M = 3000) 2.77 ms here vs 2.60 on the 64 x 96 tile, 3 windows
    // (M = 4500) 3.10 vs 3.64, 12 windows 7.75 vs 11.8
    // round 6: the cross-K/V GEMM of ONE window (N = 2 d_model x decoder layers: 18432 columns for Whisper-small = 6 x 72 tiles of 256 x 256) has the
    // tile count this form wants at M = 1500 too — the layer GEMMs (N <= 4 d_model: 54-72 tiles on 256 CUs) do not. Encoder of one window
    // 1.485 -> 1.450 ms (small.en), 7.22 -> 7.07 ms (large-v3), same bits (profiles/r6k_cross_kv_gemm3_ab.txt; WLX_CKV_GEMM3 was that A/B's switch)
    if (mode != 2 && p.mode == GEMM_CROSS_KV && p.M >= 1024 && p.N >= 8192) return true;
    return mode == 2 ? p.M >= 256 : p.M >= 4000;
}
static void gemm3_go(const GemmParams& p0, hipStream_t s) {
    GemmParams p = p0;
    if (p.rows_per_item <= 0) p.rows_per_item = 4;
    const bool scatter = p.mode == GEMM_QKV || p.mode == GEMM_CROSS_KV;
    p.epi_lds = ((p.rows_per_item & 3) == 0 && (!scatter || p.d % 256 == 0)) ? 1 : 0;       // LDS-transposed epilogue wherever the shape allows it
    p.xcd_a = 0;
    p.g3_gx = p.N / 256;
    p.g3_tiles = p.g3_gx * ((p.M + 255) / 256);
    // one workgroup per CU (the kernel holds 128 KiB of LDS and 256 registers); fewer when there are fewer tiles. A multiple of 8:
    // workgroup id % 8 is the XCD
    static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount; return n; }();
    const int nwg = std::min((n_cu / 8) * 8, ((p.g3_tiles + 7) / 8) * 8);   // (workgroups past the end of their XCD's run leave at once)
    hipLaunchKernelGGL(gemm3_kernel, dim3(nwg), dim3(512), G3_LDS_BYTES, s, p);
}

// the second form's launch: (WNT, WMT) wave tiles -> workgroup tile (32 WNT) x (32 WMT), ring of DEPTH stages, XCD-aware tile map
template <int WNT, int WMT, int DEPTH, int KS = 2>
static void gemm2_go(const GemmParams& p0, int zbatch, hipStream_t s) {
    constexpr size_t shm = (size_t)DEPTH * (2 * KS * (WNT + WMT)) * 1024;
    const int NT_total = (p0.N + 15) / 16;
    dim3 grid((NT_total + 2 * WNT - 1) / (2 * WNT), (p0.M + 32 * WMT - 1) / (32 * WMT), zbatch);
    GemmParams p = p0;
    p.xcd_a = p.xcd_b = 0;
    p.epi_lds = 1;                                          // fp16 outputs leave through the LDS-transposed epilogue
    if (p.rows_per_item <= 0) p.rows_per_item = 4;          // (modes without items: only its divisibility is looked at)
    const int gx = (int)grid.x, gy = (int)grid.y;
    if (zbatch == 1 && (gx * gy) % 8 == 0 && gx * gy >= 16) {
        // split of the 8 XCDs into a m-parts x b n-parts that minimises the bytes one XCD's L2 must hold:
        // activations / a + weights / b (both x K x 2 bytes; K cancels)
        double best = 1e300;
        for (int a = 1; a <= 8; a <<= 1) {
            const int b = 8 / a;
            if (gy % a || gx % b) continue;
            const double bytes = (double)p.M / a + (double)p.N / b;
            if (bytes < best) { best = bytes; p.xcd_a = a; p.xcd_b = b; }
        }
    }
    hipLaunchKernelGGL((gemm2_kernel<WNT, WMT, DEPTH, KS>), grid, dim3(256), shm, s, p);
}
template <int WNT, int WMT, int DEPTH, int KS = 2>
static hipError_t gemm2_optin() {      // > 64 KiB of dynamic LDS needs the opt-in, per device (not a stream operation)
    constexpr size_t shm = (size_t)DEPTH * (2 * KS * (WNT + WMT)) * 1024;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<WNT, WMT, DEPTH, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
}
// Tile shape. Measured on the Whisper-small encoder (MI355X, one shape forced for every GEMM, profiles/r2c_*): 64 x 96
// workgroup tiles (2 x 3 wave tiles, ring of 4) 1.87 ms per encoder, 64 x 64 1.97, 128 x 64 2.00, 64 x 128 2.14,
// 128 x 128 2.25, 128 x 192 2.63, and the per-GEMM "fewest rounds" choice of the first draft 1.92: the stage fills are
// latency-bound (SQ_WAIT_ANY 41-43 % of wave cycles), so what pays is MORE workgroups per CU (two 80 KiB rings fit the
// 160 KiB of LDS, i.e. 120 KiB of fills in flight per CU), not fewer, larger tiles. WLX_GEMM2_SHAPE=i forces entry i.
// Measured later in round 2 and NOT adopted (encoder-only runs, profiles/r2q / r2v): a ring of 7 stages (140 KiB) for the
// launches with <= 1 workgroup per CU (1.65 vs 1.63 ms), every LDS fragment read of a stage issued before its first MFMA
// (sched_barrier; 1.62-1.65 vs 1.61-1.63), 64 x 64 or 64 x 128 tiles for the N = d_model GEMMs (1.71-1.75 / 1.73), four k-tiles per stage
// with a ring of 3 (half the barriers: 1.74): neither the fill latency, nor the LDS read latency, nor the barrier count of a
// stage is what its 0.7 us (12 MFMAs = 0.08 us) are spent on. Phase-removal probes on this kernel (timing only, encoder-only
// runs, 1.60 ms baseline): every fill re-reading stage 0's cache-hot lines 1.61 (so not L2 / HBM), NO fills after the prologue
// 1.33, no MFMAs 1.53. What is left, and scales with co-resident workgroups per CU (36 workgroup-stages per CU take
// 0.77 us each whether one or two workgroups share the CU), is the CU's LDS port: a stage moves 20 KiB in by LDS-DMA and
// 40 KiB out through ds_read_b128 (five 1 KiB fragment reads per six MFMAs per wave). Next step for this kernel: weight
// fragments straight from their packed global image into registers (inline-asm loads with hand-counted vmcnt beside the
// LDS-DMA ring), activations only through LDS: -40 % LDS traffic per stage.
// -> Built and measured (a third form: inline-asm global_load_dwordx4 into three rotating register sets, loop unrolled by
// the ring depth, both queues counted by hand; bit-identical results): 1.586-1.591 ms vs 1.583-1.598 — no change either, so the
// LDS port is not it; the kernel was dropped again. What the probes leave standing: ~0.45 us of a 0.64 us stage remain with
// neither fills nor MFMAs, i.e. the per-stage wait + barrier + LDS-read + issue sequence of four lock-stepped waves itself.
int gemm_prepare_device() {      // once per engine, on the engine's device (wlx_engine_create)
    hipError_t e = gemm2_optin<2, 3, 4>();
    if (e == hipSuccess) e = gemm2_optin<3, 3, 3>();
    if (e == hipSuccess) e = gemm2_optin<4, 4, 2>();
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS_BYTES);
    return (int)e;
}

// Tile shape of the second form, per GEMM (round 5; cost model re-fitted in round 6). One window (M = 1500) on the 64 x 96 tile alone: the N = 3 d / 4 d
// projections are 576 / 768 workgroups against the 512 that are resident together (two 64-80 KiB rings per CU x 256 CUs), i.e. TWO rounds of a
// latency-bound stage loop. Three shapes that all keep two workgroups per CU are instantiated — 64 x 96 (ring of 4, 80 KiB), 96 x 96 (ring of 3,
// 72 KiB), 128 x 128 (ring of 2, 64 KiB). Results do not depend on the shape (same products, same K order).
//   The cost model (round 6, profiles/r6p_gemm2_shapes_by_grid.txt: every shape forced for every GEMM of small.en and large-v3, after the
// full-line activation pieces): a launch takes (stages) x (time of one stage of the shape with ONE workgroup on the CU: 0.42 / 0.50 / 0.86 us) x
// (occupancy), where every full round of 2 x CUs workgroups counts h2 = 1.43 / 1.45 / 1.17 — two co-resident workgroups share the CU's fill and LDS
// paths, so they take almost twice one workgroup's time on the small tiles and overlap better on the MFMA-denser 128 x 128 — and the last, partial
// round 1 (<= one workgroup per CU) or h2. The round-5 model (rounds of 2 x CUs, equal weight) put large-v3's N = d projections (320 workgroups)
// on 64 x 96 at 34 us where 96 x 96 (224 workgroups, one per CU) takes 27 us. All six measured (model, GEMM) cases now get their fastest shape.
// WLX_GEMM2_SHAPE=0/1/2 (A/B builds) forces one shape for every launch.
static int gemm2_pick(const GemmParams& p, int zbatch) {
    static const int forced = [] { const char* e = wlx_ab("WLX_GEMM2_SHAPE"); return e ? atoi(e) : -1; }();
    if (forced >= 0 && forced <= 2) return forced;
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount; return n; }();
    static const int wnt[3] = {2, 3, 4}, wmt[3] = {3, 3, 4};
    static const double t1[3] = {0.42, 0.50, 0.86}, h2[3] = {1.43, 1.45, 1.17}, fixed[3] = {0.0, 0.0, 1.5};
    const bool scatter = p.mode == GEMM_QKV || p.mode == GEMM_CROSS_KV;
    const double stages = (double)(p.KT / 2);
    int best = 0;
    double best_cost = 1e300;
    for (int i = 0; i < 3; ++i) {
        if (scatter && p.d % (32 * wnt[i]) != 0) continue;             // the LDS-transposed epilogue needs q / k / v boundaries on tile boundaries
        const long wgs = (long)((p.N + 32 * wnt[i] - 1) / (32 * wnt[i])) * ((p.M + 32 * wmt[i] - 1) / (32 * wmt[i])) * zbatch;
        const long full = wgs / (2 * cus), rem = wgs - full * 2 * cus;
        const double occ = (double)full * h2[i] + (rem == 0 ? 0.0 : rem <= cus ? 1.0 : h2[i]);
        const double cost = fixed[i] + stages * t1[i] * occ;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = i; }
    }
    return best;
}

// Every encoder GEMM: the large-M form where it applies (batched encodes), else the second form on the tile gemm2_pick chooses (the first
// form and the other five tile shapes measured in rounds 2-4 left the library in round 5).
void launch_gemm(const GemmParams& p, int zbatch, hipStream_t s) {
    if (gemm3_ok(p, zbatch)) { gemm3_go(p, s); return; }
    if ((p.KT & 1) || p.KT < 2) {      // every K the engine builds is a multiple of 64 (d_model, ffn: multiples of 128; conv1: 3 n_mels padded)
        fprintf(stderr, "[wlx] encoder GEMM with K = %d refused: the k-tile count must be even (wlx_engine_create validates the model shape)\n", p.KT * 32);
        return;
    }
    switch (gemm2_pick(p, zbatch)) {
        case 1: gemm2_go<3, 3, 3>(p, zbatch, s); return;
        case 2: gemm2_go<4, 4, 2>(p, zbatch, s); return;
        default: gemm2_go<2, 3, 4>(p, zbatch, s); return;
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
    float4 gv[8], bv[8];                 // gamma / beta requested WITH the row (round 6): they were loaded in the last loop, a second dependent
    const float4* g4 = reinterpret_cast<const float4*>(gamma);      // round trip behind the two reductions, in a kernel that is one trip long
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int j = lane + i * 64;
        if (j < n4) { v[i] = xr[j]; gv[i] = g4[j]; bv[i] = b4[j]; }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int j = lane + i * 64;
        if (j < n4) s += v[i].x + v[i].y + v[i].z + v[i].w;
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
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int j = lane + i * 64;
        if (j < n4) {
            const float4 gg = gv[i], bb = bv[i];
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
