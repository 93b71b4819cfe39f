// kernels.h — host-callable launchers of the gfx950 kernels (one .hip file per group).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace wlx {

// ---------------------------------------------------------------- pack.hip
// W[n][k] fp32 (row stride ldw) -> packed fp16 fragments at n-tile offset nt0 of a [NT_total][KT] image.
void launch_pack_linear(const float* W, int N, int K, long ldw, half_t* Wp, int KT, int nt0, hipStream_t s);
// conv weight W[c][ci][3] fp32 -> rows n=c, k = j*Cin + ci
void launch_pack_conv3(const float* W, int Cout, int Cin, half_t* Wp, int KT, hipStream_t s);
void launch_f32_to_f16(const float* src, half_t* dst, long n, hipStream_t s);

// ---------------------------------------------------------------- logmel.hip
struct LogmelConsts {          // device pointers, built once per engine
    const float* window;       // [400] periodic Hann
    const float* twiddle;      // [400][2] cos, sin of 2*pi*j/400 (float64-rounded)
    const float* filters;      // [n_mels][201] Slaney mel filterbank
    const int*   frange;       // [n_mels][2] first/last+1 non-zero bin
};
// up to WLX_LM_MAXB items per launch (blockIdx.y = item): per item the PCM, its length, the feature matrix, T = (n+160)/160 and a uint scratch
#define WLX_LM_MAXB 16
// An item's samples are pcm[0 .. n), or — round 6, the PCM ring — the CONCATENATION of nr ranges of `pcm`: rng = device table of
// nr x {first sample of the range in pcm, position of that sample in the concatenation} (int64 pairs, positions ascending from 0);
// the loads that bring a tile's samples into LDS walk the table (a tile of 8 frames spans at most a few ranges).
#define WLX_LM_MAXRANGES 256
struct LogmelBatch {
    int n_items;
    const float* pcm[WLX_LM_MAXB]; long n[WLX_LM_MAXB]; float* feats[WLX_LM_MAXB]; int T[WLX_LM_MAXB]; unsigned* gmax[WLX_LM_MAXB];
    const long long* rng[WLX_LM_MAXB]; int nr[WLX_LM_MAXB];
};
void launch_logmel_batch(const LogmelBatch& lb, int n_mels, const LogmelConsts& c, long ld, hipStream_t s);
// pcm [n] f32 device -> feats [n_mels][ld] f32 device (T = (n+160)/160 columns valid); gmax: 1 uint scratch
void launch_logmel(const float* pcm, long n, int n_mels, const LogmelConsts& c, float* feats, long ld,
                   int T, unsigned* gmax, hipStream_t s);
// feats window [seek, seek+seg) -> time-major fp16 featT rows 1..3000 (row stride n_mels), zero beyond seg
struct PrepWindows { int seek[64]; int seg[64]; };     // per item of a slot (wlx_slot_create: max_batch <= 64)
void launch_prep_windows(const float* feats0, long ld, int n_mels, long item_stride, const PrepWindows& w, int items,
                         half_t* featT0, long featT_stride, hipStream_t s);

// ---------------------------------------------------------------- gemm.hip
enum GemmMode : int {
    GEMM_STORE_F16 = 0,    // C[m][n] = f16(acc + bias)
    GEMM_GELU_F16 = 1,     // C[m][n] = f16(gelu(acc + bias))
    GEMM_GELU_POS_F32 = 2, // X[m][n] = gelu(acc + bias) + pos[m][n]          (conv2 + positions)
    GEMM_RESID_F32 = 3,    // X[m][n] += acc + bias                            (residual stream)
    GEMM_QKV = 4,          // n<d: q (scaled) ; d<=n<2d: k ; else v transposed
    GEMM_CROSS_KV = 5      // per decoder layer l: k rows, v transposed
};
struct GemmParams {
    const half_t* A; long lda; long strideA;   // activations, row-major fp16; z-batch stride
    const half_t* Wp; int KT;                  // packed weights
    int M, N;                                  // rows per z-batch, real output columns
    int mode;
    const float* bias;                         // [N] or null
    half_t* C; long ldc; long strideC;         // fp16 output (modes 0,1; q for mode 4)
    float* X; long ldx; long strideX;          // fp32 output (modes 2,3)
    const float* pos;                          // [M][N] (mode 2)
    // mode 4/5 extras
    int d;                                     // d_model
    float qscale;
    half_t* Kout; long ldk;                    // k rows  [rows][d]
    half_t* Vt; long ldvt;                     // v transposed [d][ldvt] per item
    int rows_per_item;                         // 1500
    long kv_item_stride_k, kv_item_stride_v;   // per item
    long kv_layer_stride_k, kv_layer_stride_v; // per decoder layer (mode 5)
    int xcd_a, xcd_b;                          // (set by the launcher, second GEMM form) XCD-aware tile map: a m-parts x b n-parts
    int epi_lds;                               // (set by the launcher, second GEMM form) fp16 outputs leave through an LDS-transposed epilogue
    int g3_gx, g3_tiles;                       // (set by the launcher, third GEMM form) tiles along n, tiles in all
};
void launch_gemm(const GemmParams& p, int zbatch, hipStream_t s);
// per-device set-up of the GEMM kernels (large dynamic-LDS opt-in); returns a hipError_t value
int gemm_prepare_device();
// x fp32 [M][d] -> fp16 LN(x)*gamma+beta [M][d]
void launch_layernorm_f16(const float* x, long ldx, const float* gamma, const float* beta,
                          half_t* out, long ldo, int M, int d, hipStream_t s);
// x fp32 [M][d] -> fp32 LN (final encoder LN, API copy) and fp16
void launch_layernorm_f16_f32(const float* x, long ldx, const float* gamma, const float* beta,
                              half_t* out16, float* out32, long ldo, int M, int d, hipStream_t s);

// ---------------------------------------------------------------- attention.hip
// non-causal encoder self-attention, head_dim 64. Q,K row-major fp16 (q pre-scaled), Vt [H*64][ldvt].
void launch_attn_encoder(const half_t* Q, long ldq, const half_t* K, long ldk, const half_t* Vt, long ldvt,
                         half_t* O, long ldo, int T, int H, int items,
                         long item_stride_q, long item_stride_k, long item_stride_vt, long item_stride_o,
                         hipStream_t s);

}  // namespace wlx
