// common.h — device-side helpers shared by the gfx950 kernels of libwlx.
// CDNA4 only: wave = 64 lanes, MFMA 16x16x32 f16 with f32 accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A/B switches (host side). The production library reads NO experiment switches from the environment: wlx_ab() is a constant there and
// every alternative behind it is dead code the optimiser drops. `-DWLX_AB` (whisperlive_amd/_lib.py build_variant("ab", ["WLX_AB"]) ->
// libwlx_ab.so, loaded with WLX_LIB) builds the library in which the surviving switches read the environment, for
// re-measuring an alternative from the same source. Runtime CONFIGURATION (WLX_SLOT_CU_MASK, WLX_DEDICATED_QUEUES, WLX_NO_GRAPH,
// WLX_QUIET, WLX_GEN_TRACE) and the two test hooks of the prompt prefill stay ordinary getenv() reads in engine.hip.
#include <cstdlib>
#ifdef WLX_AB
static inline const char* wlx_ab(const char* name) { return getenv(name); }
#else
static inline const char* wlx_ab(const char*) { return nullptr; }
#endif

#define WLX_WAVE 64
#define WLX_T_AUDIO 1500      // encoder positions per 30 s window
#define WLX_T_AUDIO_PAD 1536  // key padding so 32-key tiles never read out of bounds
#define WLX_N_FRAMES 3000     // mel frames per window
#define WLX_T_TEXT 448
#define WLX_HEAD_DIM 64

// MFMA fragment conventions (v_mfma_f32_16x16x32_f16), lane l, c = l & 15, g = l >> 4:
//   A operand: A[i = c][k = g*8 + e], e = 0..7      (8 halfs = 16 B per lane)
//   B operand: B[k = g*8 + e][j = c]
//   C/D     : D[i = g*4 + r][j = c],  r = 0..3
// Weights W[N][K] are stored PACKED per (n-tile, k-tile): Wp[((nt*KT + kt)*64 + l)*8 + e] =
//   W[nt*16 + c][kt*32 + g*8 + e], so one wave-load of a fragment is one contiguous 1 KiB.
// All GEMMs use the "swapped" form D[i = n][j = m] = sum_k W[n][k] * X[m][k]: the weight
// fragment is the A operand, the activation fragment the B operand, and every lane ends up
// with 4 CONSECUTIVE output columns n of one activation row m (wide row-major stores).

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

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
// the same exchange as a sum: the two additions are those of `v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);` with the operands of some
// rows swapped — fp32 addition commutes, so the result is bit-identical to that form, without its two ds_bpermute round trips
__device__ __forceinline__ float rows4_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a = a + b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// erf GELU — Whisper's activation (HF configuration_whisper.py: activation_function="gelu", the erf form, not tanh).
// erf by Abramowitz-Stegun 7.1.26 evaluated in fp32: |error| <= 6.1e-7 over the reals (checked against scipy on 2M
// points; torch's own fp32 GELU is 1.2e-6 from the exact value), i.e. far below the fp16 the result is rounded to.
// libm's erff inlines to ~1.4 KiB of code per call site: 16 call sites per lane in the encoder GEMM epilogue (77 KiB
// kernel) and ~0.5 us of cold instruction fetch on the decode step's fc1 launch; this is ~15 instructions.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float p = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float y = 1.0f - p * __expf(-ax * ax);
    return __builtin_copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}

__device__ __forceinline__ f16x8 ld_f16x8(const half_t* p) {
    return *reinterpret_cast<const f16x8*>(p);
}
// streamed-once data (decode weights): non-temporal, measured 0.3 us per launch better than the default policy
__device__ __forceinline__ f16x8 ld_nt_f16x8(const half_t* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p));
}
__device__ __forceinline__ f16x4 ld_f16x4(const half_t* p) {
    return *reinterpret_cast<const f16x4*>(p);
}

#define WLX_NEG_INF (-__builtin_inff())

// ------------------------------------------------------------------ DPP wave reductions (gfx9 row_bcast forms)
// One VALU instruction per butterfly step: v_<op>_dpp dst, dpp(src0), src1. hipcc does not form these from
// __builtin_amdgcn_update_dpp + fmaxf/min (it emits v_mov, v_mov_dpp, a canonicalising v_max and the v_max: 5 per step),
// so they are written out; the s_nop 1 before every step is the 2-wait-state DPP read-after-VALU-write hazard that the
// compiler does not insert inside an asm statement. After the last step lane 63 holds the wave result.
#define WLX_DPP_REDUCE(OP, V)                                                                   \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"    \
                 "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"    \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"        \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"             \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"           \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"           \
                 "s_nop 1" : "+v"(V))
#define WLX_DPP_REDUCE_ROW(OP, V)   /* each 16-lane row separately: every lane of a row ends with the row's result */ \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"    \
                 "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"    \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"        \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"             \
                 "s_nop 1" : "+v"(V))
// eight independent sums over each aligned group of 8 lanes (butterfly: every lane of a group ends with its group's total, the same bits in all
// eight — the adds of a step are commutative pairs). The eight chains are interleaved, so a register is read by DPP seven instructions after it
// was written: only the first step needs the hazard nop.
#define WLX_DPP_SUM8x8(V)                                                                                            \
    asm volatile("s_nop 1\n\t"                                                                                       \
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                           \
                 "s_nop 1"                                                                                           \
                 : "+v"((V)[0]), "+v"((V)[1]), "+v"((V)[2]), "+v"((V)[3]), "+v"((V)[4]), "+v"((V)[5]), "+v"((V)[6]), "+v"((V)[7]))
__device__ __forceinline__ float dpp_wave_sum(float v) {      // wave-uniform result
    WLX_DPP_REDUCE("v_add_f32_dpp", v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float dpp_wave_max(float v) {
    WLX_DPP_REDUCE("v_max_f32_dpp", v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int dpp_wave_min_i32(int v) {
    WLX_DPP_REDUCE("v_min_i32_dpp", v);
    return __builtin_amdgcn_readlane(v, 63);
}

// ------------------------------------------------------------------ WLX_TRACE: in-kernel timeline (profiling builds only)
// libwlx_trace.so (scripts/trace_step.py) is the same source compiled with -DWLX_TRACE: wave 0 of every workgroup of
// the decode-step kernels stamps s_memrealtime (100 MHz, chip-wide) at entry/exit (plain stores, no atomics: 3 atomics per workgroup on one
// header word were measured to add up to 7 us of serialised tail to a 192-workgroup launch) and s_memtime (shader clock) at a
// few interior marks; the marks drain outstanding memory operations first, so a traced kernel is slightly slower
// than the production one — the trace tells WHERE a launch spends its time, the production build how long it takes.
#ifdef WLX_TRACE
#define WLX_TR_MAXWG 2048
#define WLX_TR_NMARK 6
#define WLX_TR_REC 8                       // u64 per workgroup record: rt0, rt1, mark[0..5]
#define WLX_TR_STRIDE ((WLX_TR_MAXWG + 1) * WLX_TR_REC)   // per launch: header record + workgroup records
struct WlxTrace { unsigned long long* buf; int seq; };
#define WLX_TR_PARAM , WlxTrace trc
#define WLX_TR_FIELD WlxTrace trc;
__device__ __forceinline__ unsigned long long wlx_rt() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ unsigned long long wlx_ct() { return __builtin_amdgcn_s_memtime(); }
#define WLX_TR_BEGIN() unsigned long long _trm[WLX_TR_NMARK] = {0, 0, 0, 0, 0, 0}; \
    const unsigned long long _trt0 = wlx_rt(); _trm[0] = wlx_ct();
#define WLX_TR_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); _trm[i] = wlx_ct(); } while (0)
#define WLX_TR_MARK_NOWAIT(i) do { _trm[i] = wlx_ct(); } while (0)
#define WLX_TR_END(TRC) do { \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    if ((TRC).buf && threadIdx.x == 0) { \
        const unsigned long long _trt1 = wlx_rt(); \
        const int _wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
        unsigned long long* _h = (TRC).buf + (size_t)(TRC).seq * WLX_TR_STRIDE; \
        if (_wg < WLX_TR_MAXWG) { \
            unsigned long long* _r = _h + (size_t)(_wg + 1) * WLX_TR_REC; \
            _r[0] = _trt0; _r[1] = _trt1; \
            for (int _i = 0; _i < WLX_TR_NMARK; ++_i) _r[2 + _i] = _trm[_i]; \
        } \
    } } while (0)
// per-WAVE records (lane 0 of every wave, workgroups 0..127, <= 16 waves each): which wave of a workgroup is late?
#define WLX_TR_END_WAVES(TRC) do { \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    if ((TRC).buf && (threadIdx.x & 63) == 0) { \
        const unsigned long long _trt1 = wlx_rt(); \
        const int _wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
        const int _ix = _wg * 16 + (threadIdx.x >> 6); \
        unsigned long long* _h = (TRC).buf + (size_t)(TRC).seq * WLX_TR_STRIDE; \
        if (_wg < 128 && _ix < WLX_TR_MAXWG) { \
            unsigned long long* _r = _h + (size_t)(_ix + 1) * WLX_TR_REC; \
            _r[0] = _trt0; _r[1] = _trt1; \
            for (int _i = 0; _i < WLX_TR_NMARK; ++_i) _r[2 + _i] = _trm[_i]; \
        } \
    } } while (0)
#else
#define WLX_TR_END_WAVES(TRC)
#define WLX_TR_PARAM
#define WLX_TR_FIELD
#define WLX_TR_BEGIN()
#define WLX_TR_MARK(i)
#define WLX_TR_MARK_NOWAIT(i)
#define WLX_TR_END(TRC)
#endif
