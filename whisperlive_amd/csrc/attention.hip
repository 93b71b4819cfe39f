// attention.hip — encoder self-attention (non-causal, 1500 x 1500, head_dim 64), flash-style.
// Part of the ctranslate2 Whisper.encode replacement
// (whisper_live/transcriber/transcriber_faster_whisper.py:1339-1348; network definition
// HF modeling_whisper.py:267-309: q scaled by head_dim^-0.5, no bias on k).
//
// One 64-lane wave owns QT*16 query rows of one head and walks the keys in tiles of 32 with an
// online softmax. The score tile is computed TRANSPOSED, S^T = K * Q^T (K rows are the MFMA A
// operand, Q^T the B operand), so each lane holds 8 keys x ONE query: the softmax row-reduction
// is in-lane plus two cross-lane shuffles (lanes l, l^16, l^32, l^48 share a query), and the
// probabilities are already in the B-operand layout of the second product O^T = V^T * P^T — no
// LDS transpose of P. V is stored transposed ([head_dim][key]) by the QKV GEMM epilogue so the
// V^T A-fragments are two 8-byte loads per lane. Everything is read straight from L2 (K/V of a
// head are 384 KB and stay resident), no LDS, no barriers; the next key tile is prefetched into registers.
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace wlx {

template <int QT>
__global__ __launch_bounds__(64) void attn_encoder_kernel(const half_t* __restrict__ Q, long ldq,
                                                          const half_t* __restrict__ K, long ldk,
                                                          const half_t* __restrict__ Vt, long ldvt,
                                                          half_t* __restrict__ O, long ldo, int T,
                                                          long isq, long isk, long isv, long iso) {
    const int lane = threadIdx.x;
    const int c = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, item = blockIdx.z;
    const int q0 = blockIdx.x * (QT * 16);
    Q += (long)item * isq + h * WLX_HEAD_DIM;
    K += (long)item * isk + h * WLX_HEAD_DIM;
    Vt += (long)item * isv + (long)h * WLX_HEAD_DIM * ldvt;
    O += (long)item * iso + h * WLX_HEAD_DIM;

    // Q^T B-fragments: lane (j = query c, k = d = kt*32 + g*8 + e)
    f16x8 qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int row = q0 + qt * 16 + c;
        if (row >= T) row = T - 1;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) qf[qt][kt] = ld_f16x8(Q + (long)row * ldq + kt * 32 + g * 8);
    }

    f32x4 acc[QT][4];
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mrun[qt] = WLX_NEG_INF;
        lrun[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    const half_t* kbase = K + (long)c * ldk + g * 8;          // + key*ldk + kt*32
    const half_t* vbase = Vt + (long)c * ldvt + g * 4;        // + dt*16*ldvt + key0 (+16)

    // K / V^T fragments of a key tile; the NEXT tile's are requested before this tile is multiplied (a lone wave per
    // workgroup has nothing else to hide the L2 round trip behind: without the prefetch every tile paid it in full,
    // 47 tiles x ~1.2 us = 56 us per layer launch)
    f16x8 kf[2][2], vf[4];
    auto load_tile = [&](int key0, f16x8 (&kd)[2][2], f16x8 (&vd)[4]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kd[s][kt] = ld_f16x8(kbase + (long)(key0 + s * 16) * ldk + kt * 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const half_t* vp = vbase + (long)dt * 16 * ldvt + key0;
            f16x4 lo = ld_f16x4(vp), hi = ld_f16x4(vp + 16);
            vd[dt] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
    load_tile(0, kf, vf);
    for (int key0 = 0; key0 < T; key0 += 32) {
        f16x8 kn[2][2], vn[4];
        load_tile((key0 + 32 < T) ? key0 + 32 : key0, kn, vn);       // (the last trip re-reads its own tile)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 st[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                st[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) st[s] = mfma16(kf[s][kt], qf[qt][kt], st[s]);
            }
            // lane holds keys key0 + s*16 + g*4 + r for query c
            float p[8];
            float tmax = WLX_NEG_INF;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int key = key0 + s * 16 + g * 4 + r;
                    float v = (key < T) ? st[s][r] : WLX_NEG_INF;
                    p[s * 4 + r] = v;
                    tmax = fmaxf(tmax, v);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(mrun[qt], tmax);
            const float alpha = __expf(mrun[qt] - mnew);
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { p[i] = __expf(p[i] - mnew); psum += p[i]; }
            lrun[qt] = lrun[qt] * alpha + psum;   // per-lane partial; reduced over g at the end
            mrun[qt] = mnew;
            f16x8 pf = {(half_t)p[0], (half_t)p[1], (half_t)p[2], (half_t)p[3],
                        (half_t)p[4], (half_t)p[5], (half_t)p[6], (half_t)p[7]};
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 a = acc[qt][dt];
                a[0] *= alpha; a[1] *= alpha; a[2] *= alpha; a[3] *= alpha;
                acc[qt][dt] = mfma16(vf[dt], pf, a);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kf[s][kt] = kn[s][kt];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vf[dt] = vn[dt];
    }

#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = lrun[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int row = q0 + qt * 16 + c;
        if (row < T) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f16x4 o = {(half_t)(acc[qt][dt][0] * inv), (half_t)(acc[qt][dt][1] * inv),
                           (half_t)(acc[qt][dt][2] * inv), (half_t)(acc[qt][dt][3] * inv)};
                *reinterpret_cast<f16x4*>(O + (long)row * ldo + dt * 16 + g * 4) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Second form (round 2). Same tiling and MFMA order as attn_encoder_kernel; what changed is what the ISA and the
// rocprofv3 SQ counters of the first form showed it was spending its time on (profiles/r2c_*):
//  * hipcc collapsed the one-tile "prefetch" above (the kn/vn -> kf/vf copies were coalesced, so the loads of tile i+1
//    land in the registers the MFMAs of tile i+1 read and every tile still waits a full L2 round trip, then drains
//    vmcnt(0) at the loop end): here three NAMED tile register sets rotate through a loop body of six tiles, two tiles
//    are always in flight behind the one being multiplied, and nothing is copied;
//  * the softmax row maximum crossed the four 16-lane rows of a wave with two ds_bpermute round trips per query tile:
//    v_permlane16_swap / v_permlane32_swap exchange rows inside the VALU;
//  * with the waiting gone the kernel is VALU-ISSUE bound (SQ_ACTIVE_INST_VALU 56 % of wave cycles, 290 VALU instructions
//    per 32-key tile): a third of them were v_accvgpr_read/write around the accumulator rescale (gone with the library-
//    wide -amdgpu-mfma-vgpr-form, _lib.py), and the rest is trimmed here — scores in the log2 domain (FMA + v_exp_f32
//    per score), key masks only in the last tiles, the rescale skipped when no lane raised its maximum: 142 per tile.
// Results differ from the first form by fp32 rounding of the exponent argument only (parity tests: unchanged error).
__device__ __forceinline__ float rows4_max(float v) {
    // rows (16-lane groups) r0..r3 of a wave: after the first swap a = {r0,r0,r2,r2}, b = {r1,r1,r3,r3}; after the second
    // a = {lo,lo}, b = {hi,hi}. Written as asm with BOTH operands read-write: the builtin with two identical operands is
    // folded by hipcc (ROCm 7.2) as if it returned its input twice, which silently drops the max.
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a = fmaxf(a, b);
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

struct AttnTile { f16x8 k[2][2]; f16x8 v[4]; };

template <int QT>
__global__ __launch_bounds__(64) void attn_encoder_pf_kernel(const half_t* __restrict__ Q, long ldq,
                                                             const half_t* __restrict__ K, long ldk,
                                                             const half_t* __restrict__ Vt, long ldvt,
                                                             half_t* __restrict__ O, long ldo, int T,
                                                             long isq, long isk, long isv, long iso) {
    const int lane = threadIdx.x;
    const int c = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, item = blockIdx.z;
    const int q0 = blockIdx.x * (QT * 16);
    Q += (long)item * isq + h * WLX_HEAD_DIM;
    K += (long)item * isk + h * WLX_HEAD_DIM;
    Vt += (long)item * isv + (long)h * WLX_HEAD_DIM * ldvt;
    O += (long)item * iso + h * WLX_HEAD_DIM;

    f16x8 qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int row = q0 + qt * 16 + c;
        if (row >= T) row = T - 1;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) qf[qt][kt] = ld_f16x8(Q + (long)row * ldq + kt * 32 + g * 8);
    }
    f32x4 acc[QT][4];
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mrun[qt] = WLX_NEG_INF;
        lrun[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const half_t* kbase = K + (long)c * ldk + g * 8;
    const half_t* vbase = Vt + (long)c * ldvt + g * 4;
    const int NT = (T + 31) >> 5;
    auto load_tile = [&](int ti, AttnTile& t) {
        const int key0 = ((ti < NT) ? ti : NT - 1) << 5;              // past the end: re-read the last tile (never multiplied)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) t.k[s][kt] = ld_f16x8(kbase + (long)(key0 + s * 16) * ldk + kt * 32);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const half_t* vp = vbase + (long)dt * 16 * ldvt + key0;
            const f16x4 lo = ld_f16x4(vp), hi = ld_f16x4(vp + 16);
            t.v[dt] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
    // Scores stay in the log2 domain: p = 2^(s * log2(e) - m2) with m2 the running maximum times log2(e) — one FMA and one
    // v_exp_f32 per score instead of subtract, multiply and exp (the kernel is VALU-issue bound: rocprofv3 SQ counters,
    // profiles/r2c_*). `masked` = the tile may hold keys >= T (the last real tile and the all-masked filler).
    constexpr float LOG2E = 1.4426950408889634f;
    auto multiply = [&](int ti, const AttnTile& t, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int key0 = ti << 5;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 st[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                st[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) st[s] = mfma16(t.k[s][kt], qf[qt][kt], st[s]);
            }
            float p[8];
            float tmax = WLX_NEG_INF;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = st[s][r];
                    if (MASKED) { const int key = key0 + s * 16 + g * 4 + r; v = (key < T) ? v : WLX_NEG_INF; }
                    p[s * 4 + r] = v;
                    tmax = fmaxf(tmax, v);
                }
            tmax = rows4_max(tmax) * LOG2E;                     // (log2(e) > 0: the maximum commutes with the scaling)
            const float mnew = fmaxf(mrun[qt], tmax);
            const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { p[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[i], LOG2E, -mnew)); psum += p[i]; }
            lrun[qt] = lrun[qt] * alpha + psum;
            mrun[qt] = mnew;
            const f16x8 pf = {(half_t)p[0], (half_t)p[1], (half_t)p[2], (half_t)p[3],
                              (half_t)p[4], (half_t)p[5], (half_t)p[6], (half_t)p[7]};
            // the running maximum of most queries settles within the first tiles: when NO lane of the wave raised its
            // maximum (alpha == 1 everywhere) the 16 accumulator multiplies are skipped — by 1.0f they change nothing
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
                float al = alpha;
                asm volatile("" : "+v"(al));                    // opaque: keeps this a wave-uniform BRANCH (hipcc if-converts it into 16 selects otherwise)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { acc[qt][dt][0] *= al; acc[qt][dt][1] *= al; acc[qt][dt][2] *= al; acc[qt][dt][3] *= al; }
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[qt][dt] = mfma16(t.v[dt], pf, acc[qt][dt]);
        }
    };
    using Plain = std::false_type;
    using Masked = std::true_type;
    // A tile is 12 vector-memory loads (4 x 16 B of K, 8 x 8 B of V^T); two tiles are kept in flight behind the one being
    // multiplied. hipcc's own wait insertion is conservative at a loop header — whatever was requested before the
    // back-edge is waited for in full at its first use after it (a three-tile loop body drained the tile requested one
    // multiply earlier on every trip) — so (1) the body covers SIX tiles and crosses the back-edge with only the oldest
    // request outstanding, (2) every multiply is preceded by an explicit counted wait (__builtin_amdgcn_s_waitcnt, which
    // the compiler's bookkeeping understands) and a scheduling barrier that keeps the MFMAs below it. Tiles past the end
    // (the body runs ceil(47 / 6) * 6 = 48 tiles) re-read the last tile with every key masked: p = exp(-inf) = 0,
    // alpha = 1, so they change nothing, bit for bit.
#define WLX_WAIT_VM(n) do { __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x0F70); __builtin_amdgcn_sched_barrier(0); } while (0)
    AttnTile tA, tB, tC;
    load_tile(0, tA);
#define WLX_SIX_TILES(TAG)                                                            \
        load_tile(ti + 1, tB);                                                        \
        load_tile(ti + 2, tC);                                                        \
        WLX_WAIT_VM(24); multiply(ti, tA, TAG{});     load_tile(ti + 3, tA);          \
        WLX_WAIT_VM(24); multiply(ti + 1, tB, TAG{}); load_tile(ti + 4, tB);          \
        WLX_WAIT_VM(24); multiply(ti + 2, tC, TAG{}); load_tile(ti + 5, tC);          \
        WLX_WAIT_VM(24); multiply(ti + 3, tA, TAG{}); load_tile(ti + 6, tA);          \
        WLX_WAIT_VM(24); multiply(ti + 4, tB, TAG{});                                 \
        WLX_WAIT_VM(12); multiply(ti + 5, tC, TAG{});
    int ti = 0;
#pragma unroll 1
    for (; (ti + 6) * 32 <= T; ti += 6) { WLX_SIX_TILES(Plain) }        // six tiles entirely below T: no key masks
#pragma unroll 1
    for (; ti < NT; ti += 6) { WLX_SIX_TILES(Masked) }                  // the rest (at most one or two trips)
#undef WLX_SIX_TILES
#undef WLX_WAIT_VM
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = lrun[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int row = q0 + qt * 16 + c;
        if (row < T) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f16x4 o = {(half_t)(acc[qt][dt][0] * inv), (half_t)(acc[qt][dt][1] * inv),
                                 (half_t)(acc[qt][dt][2] * inv), (half_t)(acc[qt][dt][3] * inv)};
                *reinterpret_cast<f16x4*>(O + (long)row * ldo + dt * 16 + g * 4) = o;
            }
        }
    }
}

void launch_attn_encoder(const half_t* Q, long ldq, const half_t* K, long ldk, const half_t* Vt, long ldvt,
                         half_t* O, long ldo, int T, int H, int items,
                         long isq, long isk, long isv, long iso, hipStream_t s) {
    constexpr int QT = 2;   // measured on Whisper-small: QT = 1 fills every SIMD but doubles the K/V re-reads from L2: 2.64 vs 2.35 ms per encoder
    dim3 grid((T + QT * 16 - 1) / (QT * 16), H, items);
    static const int form = [] { const char* e = getenv("WLX_ENC_ATTN"); return e ? atoi(e) : 2; }();   // 1 = first form (A/B)
    if (form == 1)
        hipLaunchKernelGGL((attn_encoder_kernel<QT>), grid, dim3(64), 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, T,
                           isq, isk, isv, iso);
    else
        hipLaunchKernelGGL((attn_encoder_pf_kernel<QT>), grid, dim3(64), 0, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, T,
                           isq, isk, isv, iso);
}

}  // namespace wlx
