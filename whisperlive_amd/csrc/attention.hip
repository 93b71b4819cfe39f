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

// (The first two forms — one wave per 16 x QT query rows reading K / V straight from L2, then the same with three rotating
// tile register sets — were the A/B references of round 2; measured, superseded by the LDS forms below and removed from the
// library in round 5: DESIGN.md "Encoder attention".)
// (rows4_max / rows4_sum: common.h)

// ------------------------------------------------------------------------------------------------------------------
// Third form (round 2): K / V^T tiles SHARED by the four waves of a workgroup through an LDS ring filled by LDS-DMA.
// What the second form still paid (rocprofv3: 52 us per layer launch, 5 % of the MFMA peak): 564 one-wave workgroups =
// 0.55 waves per SIMD, every wave pulling its own copy of every K / V tile from L2 (212 MB of L2 -> L1 traffic per layer)
// and nothing on its SIMD to overlap its softmax with. Here a workgroup owns 64 query rows of a head (4 waves x 16 rows:
// twice the waves, so ~1.1 per SIMD and two workgroups on the CUs that get two), a tile of 32 keys is fetched ONCE per
// workgroup — eight 1 KiB global_load_lds_dwordx4 pieces, two per wave, DEPTH - 1 tiles in flight — and every wave reads
// its MFMA operands from LDS (lane-linear 16 B, conflict-free), so L2 traffic drops 4x and the loads of tile i + 3 run
// under the arithmetic of tile i. One counted vmcnt + one raw s_barrier per tile (no __syncthreads: hipcc lowers it to
// vmcnt(0) with LDS-DMA in flight and the ring would drain every trip, see gemm.hip).
//   Key order inside a tile: K fragment s (of 2) holds keys g8*8 + s*4 + r  (g8 = 0..3, r = 0..3) in its 16 A-rows, so
// the transposed score tile leaves lane (query c, g) with keys g*8 + s*4 + r — i.e. its 8 probabilities are the 8
// CONSECUTIVE keys g*8 .. g*8+7, exactly one 16-byte piece of a V^T row: V^T fragments become single dwordx4 LDS-DMA
// pieces (the second form needed two 8-byte loads per fragment, which LDS-DMA cannot express).
typedef __attribute__((address_space(3))) void wlx_lds_void_a;

// XCD-aware workgroup -> (query block, head, item) map of the LDS forms (round 4). Workgroup ids are dealt round-robin over the 8 XCDs, each
// with its own 4 MiB L2; with the query block as the fastest grid index the 24 workgroups that share one (head, item)'s K / V (384 KiB) were
// spread over all eight L2s, and at 12 windows every L2 saw the K / V of all ~32 (head, item) pairs in flight: 26 % of the L2 requests missed
// (TCC counters, profiles/r4g_*: 520 MB per launch from beyond L2 for 55 MB of K / V). Here XCD x owns the contiguous range
// [x * chunk, (x + 1) * chunk) of the (pair, query block) list — its workgroups, dispatched in order, share ~4 pairs (1.5 MiB) at a time.
struct AttnMap { int nq, H, total, chunk; };               // query blocks per pair; heads; pairs * nq; ceil(total / 8), or 0 = plain order
__device__ __forceinline__ bool attn_map(const AttnMap& am, int& qb, int& h, int& item) {
    const int L = (int)blockIdx.x;
    const int G = am.chunk > 0 ? (L & 7) * am.chunk + (L >> 3) : L;
    if (G >= am.total || (am.chunk > 0 && (L >> 3) >= am.chunk)) return false;
    const int pair = G / am.nq;
    qb = G - pair * am.nq;
    item = pair / am.H;
    h = pair - item * am.H;
    return true;
}

template <int DEPTH, int NW>
__global__ __launch_bounds__(NW * 64) void attn_encoder_lds_kernel(const half_t* __restrict__ Q, long ldq,
                                                               const half_t* __restrict__ K, long ldk,
                                                               const half_t* __restrict__ Vt, long ldvt,
                                                               half_t* __restrict__ O, long ldo, int T,
                                                               long isq, long isk, long isv, long iso, AttnMap am) {
    constexpr int NL = 8 / NW;                              // LDS-DMA pieces (vmcnt events) per wave per tile
    static_assert((NW == 4 || NW == 8) && DEPTH >= 3 && NL * (DEPTH - 2) <= 63, "8 pieces per tile over 4 or 8 waves; vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) f16x8 aring[];      // [DEPTH][8 fragments][64 lanes] x 16 B — the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    int qb, h, item;
    if (!attn_map(am, qb, h, item)) return;                 // (whole workgroup, before any barrier)
    const int q0 = qb * (NW * 16) + wave * 16;
    Q += (long)item * isq + h * WLX_HEAD_DIM;
    K += (long)item * isk + h * WLX_HEAD_DIM;
    Vt += (long)item * isv + (long)h * WLX_HEAD_DIM * ldvt;
    O += (long)item * iso + h * WLX_HEAD_DIM;
    const int NT = (T + 31) >> 5;

    // this wave's pieces of a tile: fragments NL * wave .. of (K s0kt0, K s0kt1, K s1kt0, K s1kt1, V dt0..3)
    const int f0 = NL * wave;                               // first fragment of this wave
    const bool frag_is_k = f0 < 4;
    const half_t* src0;
    long step;                                              // halfs per tile
    long frag_step;                                         // halfs between this wave's two fragments (NL == 2)
    const int ks = (f0 >> 1) & 1, kkt = f0 & 1;             // K fragment (s, kt) of piece 0
    if (frag_is_k) {
        const int krow = (c >> 2) * 8 + ks * 4 + (c & 3);   // rows in the permuted key order
        src0 = K + (long)krow * ldk + kkt * 32 + g * 8;
        step = 32 * ldk; frag_step = 32;
    } else {
        const int dt = f0 - 4;                              // V^T fragment dt: 8 consecutive keys of dim row dt*16 + c
        src0 = Vt + (long)(dt * 16 + c) * ldvt + g * 8;
        step = 32; frag_step = 16 * ldvt;
    }
    const long kclamp = (long)(T - 1) * ldk;                // K rows past the end re-read the last row (their scores are masked)
    auto issue = [&](int ti, int buf) {
        f16x8* dst = aring + ((long)buf * 8 + f0) * 64;
        const half_t* a0 = src0 + (long)ti * step;
        if (frag_is_k) {                                          // (wave-uniform) only the last tile can run past row T - 1
            const long row = (long)ti * 32 + (c >> 2) * 8 + ks * 4 + (c & 3);
            if (row >= T) a0 = K + kclamp + kkt * 32 + g * 8;
        }
        __builtin_amdgcn_global_load_lds((const void*)a0, (wlx_lds_void_a*)dst, 16, 0, 0);
        if constexpr (NL == 2) __builtin_amdgcn_global_load_lds((const void*)(a0 + frag_step), (wlx_lds_void_a*)(dst + 64), 16, 0, 0);
    };
    // The query fragments go through LDS as well (LDS-DMA into a per-wave 2 KiB area behind the ring, FIRST so that they
    // are the oldest requests): with an ordinary VGPR-destination load anywhere near an LDS-DMA pipeline hipcc's wait
    // insertion falls back to vmcnt(0) at the first use of that register on EVERY trip of the loop — the whole ring drained
    // per tile, one L2 round trip per tile (measured: 46 us per layer launch whatever the ring depth; an explicit counted
    // wait in front of the loop did not change its mind). All-LDS-DMA kernels get counted waits only.
    f16x8* qarea = aring + (long)DEPTH * 8 * 64 + wave * 2 * 64;
    {
        int row = q0 + c;
        if (row >= T) row = T - 1;
        const half_t* qsrc = Q + (long)row * ldq + g * 8;
        __builtin_amdgcn_global_load_lds((const void*)qsrc, (wlx_lds_void_a*)qarea, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(qsrc + 32), (wlx_lds_void_a*)(qarea + 64), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i)
        if (i < NT) issue(i, i);
    // the two Q pieces are the oldest requests: the DEPTH - 1 fills behind them stay in flight (NT >= DEPTH - 1 always)
    __builtin_amdgcn_s_waitcnt(((NL * (DEPTH - 1)) & 15) | (((NL * (DEPTH - 1)) >> 4) << 14) | 0x0F70);
    f16x8 qf[2];
    qf[0] = qarea[lane];
    qf[1] = qarea[64 + lane];
    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = WLX_NEG_INF, lrun = 0.f;
    constexpr float LOG2E = 1.4426950408889634f;

    int rbuf = 0, wbuf = DEPTH - 1;
    auto tile = [&](int ti, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        // tiles ti .. min(ti + DEPTH - 2, NT - 1) are outstanding; tile ti must have landed (this wave's pieces: counted
        // vmcnt; everyone's: the barrier, which also says everyone is done reading the buffer refilled next)
        if (ti + DEPTH - 2 < NT) __builtin_amdgcn_s_waitcnt(((NL * (DEPTH - 2)) & 15) | (((NL * (DEPTH - 2)) >> 4) << 14) | 0x0F70);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        if (ti + DEPTH - 1 < NT) issue(ti + DEPTH - 1, wbuf);
        const f16x8* src = aring + (long)rbuf * 8 * 64 + lane;
        // all eight fragments of the tile are requested from LDS at once: the V^T reads then land under the score MFMAs and
        // the softmax instead of costing a second exposed LDS round trip per tile (one wave per SIMD: nothing else hides it;
        // SQ counters before: 47 % of the wave cycles parked at waits, profiles/r2z_pmc_sq_encoder.csv)
        f16x8 kfr[4], vfr[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) kfr[f] = src[f * 64];
#pragma unroll
        for (int f = 0; f < 4; ++f) vfr[f] = src[(4 + f) * 64];
        __builtin_amdgcn_sched_barrier(0);
        f32x4 st[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            st[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) st[s] = mfma16(kfr[s * 2 + kt], qf[kt], st[s]);
        }
        float p[8];
        float tmax = WLX_NEG_INF;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = st[s][r];
                if (MASKED) { const int key = ti * 32 + g * 8 + s * 4 + r; v = (key < T) ? v : WLX_NEG_INF; }
                p[s * 4 + r] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = rows4_max(tmax) * LOG2E;
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { p[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[i], LOG2E, -mnew)); psum += p[i]; }
        lrun = lrun * alpha + psum;
        mrun = mnew;
        const f16x8 pf = {(half_t)p[0], (half_t)p[1], (half_t)p[2], (half_t)p[3],
                          (half_t)p[4], (half_t)p[5], (half_t)p[6], (half_t)p[7]};
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
            float al = alpha;
            asm volatile("" : "+v"(al));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { acc[dt][0] *= al; acc[dt][1] *= al; acc[dt][2] *= al; acc[dt][3] *= al; }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = mfma16(vfr[dt], pf, acc[dt]);
        rbuf = (rbuf + 1 == DEPTH) ? 0 : rbuf + 1;
        wbuf = (wbuf + 1 == DEPTH) ? 0 : wbuf + 1;
    };
    int ti = 0;
#pragma unroll 1
    for (; (ti + 1) * 32 <= T; ++ti) tile(ti, std::false_type{});          // tiles entirely below T: no key masks
#pragma unroll 1
    for (; ti < NT; ++ti) tile(ti, std::true_type{});                       // the last tile (keys >= T masked)
    float l = lrun;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int row = q0 + c;
    if (row < T) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f16x4 o = {(half_t)(acc[dt][0] * inv), (half_t)(acc[dt][1] * inv),
                             (half_t)(acc[dt][2] * inv), (half_t)(acc[dt][3] * inv)};
            *reinterpret_cast<f16x4*>(O + (long)row * ldo + dt * 16 + g * 4) = o;
        }
    }
}

// Fourth form (round 4): the third form's ring, SOFTWARE-PIPELINED inside the wave. The third form's tile is one dependent chain per wave —
// barrier -> 8 LDS reads -> 4 score MFMAs -> softmax (~55 VALU + 9 quarter-rate exponentials) -> 4 output MFMAs — and with ~1 wave per SIMD
// resident (SQ counters at 12 windows, profiles/r4g_pmc_sq_encoder_b12.csv: waves issue in 24 % of their cycles and sit at waits in 51 %) nothing
// fills the LDS and MFMA latencies: 740 cycles per 16-query x 32-key tile against ~330 cycles of VALU issue. Here a wave keeps TWO tiles in
// flight: iteration ti requests tile ti+1's K / V fragments from LDS, runs tile ti's softmax on the scores computed one iteration earlier
// (the LDS latency passes under it), then issues tile ti+1's score MFMAs and tile ti's output MFMAs back to back — they execute under the next
// iteration's barrier, LDS requests and the first softmax instructions. Same instructions per tile, same operand order: results are
// bit-identical to the third form's. Two register sets (scores 8, V fragments 16) alternate, the loop is unrolled by two.
template <int DEPTH, int NW>
__global__ __launch_bounds__(NW * 64) void attn_encoder_lds2_kernel(const half_t* __restrict__ Q, long ldq,
                                                                const half_t* __restrict__ K, long ldk,
                                                                const half_t* __restrict__ Vt, long ldvt,
                                                                half_t* __restrict__ O, long ldo, int T,
                                                                long isq, long isk, long isv, long iso, AttnMap am) {
    constexpr int NL = 8 / NW;
    static_assert((NW == 4 || NW == 8) && DEPTH >= 3 && NL * (DEPTH - 2) <= 63, "8 pieces per tile over 4 or 8 waves; vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) f16x8 aring[];      // [DEPTH][8 fragments][64 lanes] x 16 B + the query area
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    int qb, h, item;
    if (!attn_map(am, qb, h, item)) return;                 // (whole workgroup, before any barrier)
    const int q0 = qb * (NW * 16) + wave * 16;
    Q += (long)item * isq + h * WLX_HEAD_DIM;
    K += (long)item * isk + h * WLX_HEAD_DIM;
    Vt += (long)item * isv + (long)h * WLX_HEAD_DIM * ldvt;
    O += (long)item * iso + h * WLX_HEAD_DIM;
    const int NT = (T + 31) >> 5;
    const int f0 = NL * wave;
    const bool frag_is_k = f0 < 4;
    const half_t* src0;
    long step_h, frag_step;
    const int ks = (f0 >> 1) & 1, kkt = f0 & 1;
    if (frag_is_k) {
        const int krow = (c >> 2) * 8 + ks * 4 + (c & 3);
        src0 = K + (long)krow * ldk + kkt * 32 + g * 8;
        step_h = 32 * ldk; frag_step = 32;
    } else {
        const int dt = f0 - 4;
        src0 = Vt + (long)(dt * 16 + c) * ldvt + g * 8;
        step_h = 32; frag_step = 16 * ldvt;
    }
    const long kclamp = (long)(T - 1) * ldk;
    auto issue = [&](int ti, int buf) {
        f16x8* dst = aring + ((long)buf * 8 + f0) * 64;
        const half_t* a0 = src0 + (long)ti * step_h;
        if (frag_is_k) {
            const long row = (long)ti * 32 + (c >> 2) * 8 + ks * 4 + (c & 3);
            if (row >= T) a0 = K + kclamp + kkt * 32 + g * 8;
        }
        __builtin_amdgcn_global_load_lds((const void*)a0, (wlx_lds_void_a*)dst, 16, 0, 0);
        if constexpr (NL == 2) __builtin_amdgcn_global_load_lds((const void*)(a0 + frag_step), (wlx_lds_void_a*)(dst + 64), 16, 0, 0);
    };
    f16x8* qarea = aring + (long)DEPTH * 8 * 64 + wave * 2 * 64;
    {
        int row = q0 + c;
        if (row >= T) row = T - 1;
        const half_t* qsrc = Q + (long)row * ldq + g * 8;
        __builtin_amdgcn_global_load_lds((const void*)qsrc, (wlx_lds_void_a*)qarea, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(qsrc + 32), (wlx_lds_void_a*)(qarea + 64), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i)
        if (i < NT) issue(i, i);
    __builtin_amdgcn_s_waitcnt(((NL * (DEPTH - 1)) & 15) | (((NL * (DEPTH - 1)) >> 4) << 14) | 0x0F70);
    f16x8 qf[2];
    qf[0] = qarea[lane];
    qf[1] = qarea[64 + lane];
    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = WLX_NEG_INF, lrun = 0.f;
    constexpr float LOG2E = 1.4426950408889634f;
    int rbuf = 0, wbuf = DEPTH - 1;

    // LDS side of tile tl: landed (this wave's pieces: counted vmcnt; everyone's: the barrier, which also says every wave has READ the
    // buffer refilled next — its reads of tile tl-1 were waited for, lgkmcnt(0), before that tile's score MFMAs), refill, fragment requests
    auto request = [&](int tl, f16x8 (&kfr)[4], f16x8 (&vfr)[4]) {
        if (tl + DEPTH - 2 < NT) __builtin_amdgcn_s_waitcnt(((NL * (DEPTH - 2)) & 15) | (((NL * (DEPTH - 2)) >> 4) << 14) | 0x0F70);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        if (tl + DEPTH - 1 < NT) issue(tl + DEPTH - 1, wbuf);
        const f16x8* src = aring + (long)rbuf * 8 * 64 + lane;
#pragma unroll
        for (int f = 0; f < 4; ++f) kfr[f] = src[f * 64];
#pragma unroll
        for (int f = 0; f < 4; ++f) vfr[f] = src[(4 + f) * 64];
        rbuf = (rbuf + 1 == DEPTH) ? 0 : rbuf + 1;
        wbuf = (wbuf + 1 == DEPTH) ? 0 : wbuf + 1;
    };
    auto scores = [&](const f16x8 (&kfr)[4], f32x4 (&st)[2]) {
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): K AND V fragments (see request: nothing of this wave may still read the ring at the next barrier)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            st[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) st[s] = mfma16(kfr[s * 2 + kt], qf[kt], st[s]);
        }
    };
    // tile ti: softmax of its scores sc, output MFMAs with its V fragments vc; in between (has_next) tile ti+1's requests and score MFMAs
    auto step = [&](int ti, f32x4 (&sc)[2], f16x8 (&vc)[4], f32x4 (&sn)[2], f16x8 (&vn)[4], auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const bool has_next = ti + 1 < NT;                  // (uniform over the workgroup: every wave meets the same barriers)
        f16x8 kfr[4];
        if (has_next) request(ti + 1, kfr, vn);
        __builtin_amdgcn_sched_barrier(0);
        float p[8];
        float tmax = WLX_NEG_INF;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = sc[s][r];
                if (MASKED) { const int key = ti * 32 + g * 8 + s * 4 + r; v = (key < T) ? v : WLX_NEG_INF; }
                p[s * 4 + r] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = rows4_max(tmax) * LOG2E;
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { p[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[i], LOG2E, -mnew)); psum += p[i]; }
        lrun = lrun * alpha + psum;
        mrun = mnew;
        const f16x8 pf = {(half_t)p[0], (half_t)p[1], (half_t)p[2], (half_t)p[3],
                          (half_t)p[4], (half_t)p[5], (half_t)p[6], (half_t)p[7]};
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
            float al = alpha;
            asm volatile("" : "+v"(al));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { acc[dt][0] *= al; acc[dt][1] *= al; acc[dt][2] *= al; acc[dt][3] *= al; }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) scores(kfr, sn);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = mfma16(vc[dt], pf, acc[dt]);
    };
    f32x4 sA[2], sB[2];
    f16x8 vA[4], vB[4];
    {
        f16x8 kfr[4];
        request(0, kfr, vA);
        scores(kfr, sA);
    }
    const int NTfull = T >> 5;                              // tiles entirely below T
    int ti = 0;
#pragma unroll 1
    for (; ti + 2 <= NTfull; ti += 2) {
        step(ti, sA, vA, sB, vB, std::false_type{});
        step(ti + 1, sB, vB, sA, vA, std::false_type{});
    }
    if (ti < NT) {                                          // at most two more: a last full tile and / or the tile that crosses T
        step(ti, sA, vA, sB, vB, std::true_type{});
        if (ti + 1 < NT) step(ti + 1, sB, vB, sA, vA, std::true_type{});
    }
    float l = lrun;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int row = q0 + c;
    if (row < T) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f16x4 o = {(half_t)(acc[dt][0] * inv), (half_t)(acc[dt][1] * inv),
                             (half_t)(acc[dt][2] * inv), (half_t)(acc[dt][3] * inv)};
            *reinterpret_cast<f16x4*>(O + (long)row * ldo + dt * 16 + g * 4) = o;
        }
    }
}

// (Measured and dropped, profiles/r2z: a fourth form with TWO key tiles per ring stage — half the waits and barriers, the same
// bytes in flight — runs at the third form's speed (1.58-1.61 ms per encoder for both), as do a ring of 7 and eight waves per
// workgroup: the 47 % of wave cycles the SQ counters show parked are not the per-tile barrier or the fill latency. A fifth
// form — two waves per 16-query block, each walking every other key tile, merged through LDS at the end: 2.2 waves per
// SIMD instead of 1.1 — is ALSO equal (1.57-1.59 ms, large-v3 7.60 vs 7.64 ms): a SIMD retires one 16-query x 32-key tile per
// ~0.64 us however many waves share it, i.e. the kernel is bound by the ~170 instructions per tile (8 MFMAs, ~100 VALU of
// which 9 quarter-rate exponentials, hazard nops), not by anything a second wave could hide.)
int attn_prepare_device() { return 0; }                     // (the ring is 32 KiB: below the default dynamic-LDS limit)

void launch_attn_encoder(const half_t* Q, long ldq, const half_t* K, long ldk, const half_t* Vt, long ldvt,
                         half_t* O, long ldo, int T, int H, int items,
                         long isq, long isk, long isv, long iso, hipStream_t s) {
    // one window = the LDS form with 4 waves per workgroup, ring of 4; batched encodes (>= 4000 rows) = the software-pipelined LDS form with
    // EIGHT waves sharing a K / V tile (12 windows 7.41 -> 7.05 ms per encoder, large-v3 x 8 24.8 -> 24.1 ms; one window 1.63 vs 1.66 ms,
    // profiles/r4attn2_*). The other ring depths / wave counts measured in rounds 2-4 are no longer instantiated.
    // XCD-aware (pair, query block) order: see attn_map.
    const bool batched = (long)items * T >= 4000;
    const int nw = batched ? 8 : 4;
    AttnMap am;
    am.nq = (T + nw * 16 - 1) / (nw * 16); am.H = H; am.total = am.nq * H * items;
    am.chunk = (am.total + 7) / 8;
    const dim3 grid(8 * am.chunk);
    if (batched)
        hipLaunchKernelGGL((attn_encoder_lds2_kernel<4, 8>), grid, dim3(8 * 64), 4 * 8 * 1024 + 8 * 2048, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, T, isq, isk, isv, iso, am);
    else
        hipLaunchKernelGGL((attn_encoder_lds_kernel<4, 4>), grid, dim3(4 * 64), 4 * 8 * 1024 + 4 * 2048, s, Q, ldq, K, ldk, Vt, ldvt, O, ldo, T, isq, isk, isv, iso, am);
}

}  // namespace wlx
