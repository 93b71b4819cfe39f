// pack.hip — one-time weight repacking into the MFMA-fragment layout (see common.h).
// Memory is laid out for the kernels, not for the framework that produced the weights: every
// wave-level fragment load in the GEMM / GEMV kernels is one contiguous, 1 KiB-aligned burst.
#include "kernels.h"

namespace wlx {

// one thread per (n-tile, k-tile, lane): writes 8 halfs
__global__ void pack_linear_kernel(const float* __restrict__ W, int N, int K, long ldw,
                                   half_t* __restrict__ Wp, int KT, int nt0, int NT) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)NT * KT * 64;
    if (idx >= total) return;
    int lane = idx & 63;
    long tile = idx >> 6;
    int kt = tile % KT;
    int nt = tile / KT;
    int n = nt * 16 + (lane & 15);
    int k0 = kt * 32 + (lane >> 4) * 8;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int k = k0 + e;
        float f = (n < N && k < K) ? W[(long)n * ldw + k] : 0.0f;
        v[e] = (half_t)f;
    }
    *reinterpret_cast<f16x8*>(Wp + (((long)(nt0 + nt) * KT + kt) * 64 + lane) * 8) = v;
}

void launch_pack_linear(const float* W, int N, int K, long ldw, half_t* Wp, int KT, int nt0, hipStream_t s) {
    int NT = (N + 15) / 16;
    long total = (long)NT * KT * 64;
    int bs = 256;
    hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, s,
                       W, N, K, ldw, Wp, KT, nt0, NT);
}

// conv weight W[c][ci][j] (j = 0..2) -> W'[c][k = j*Cin + ci]
__global__ void pack_conv3_kernel(const float* __restrict__ W, int Cout, int Cin,
                                  half_t* __restrict__ Wp, int KT, int NT) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)NT * KT * 64;
    if (idx >= total) return;
    int lane = idx & 63;
    long tile = idx >> 6;
    int kt = tile % KT;
    int nt = tile / KT;
    int n = nt * 16 + (lane & 15);
    int k0 = kt * 32 + (lane >> 4) * 8;
    int K = 3 * Cin;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int k = k0 + e;
        float f = 0.0f;
        if (n < Cout && k < K) {
            int j = k / Cin, ci = k - j * Cin;
            f = W[((long)n * Cin + ci) * 3 + j];
        }
        v[e] = (half_t)f;
    }
    *reinterpret_cast<f16x8*>(Wp + (((long)nt * KT + kt) * 64 + lane) * 8) = v;
}

void launch_pack_conv3(const float* W, int Cout, int Cin, half_t* Wp, int KT, hipStream_t s) {
    int NT = (Cout + 15) / 16;
    long total = (long)NT * KT * 64;
    int bs = 256;
    hipLaunchKernelGGL(pack_conv3_kernel, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, s,
                       W, Cout, Cin, Wp, KT, NT);
}

__global__ void f32_to_f16_kernel(const float* __restrict__ src, half_t* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (half_t)src[i];
}

void launch_f32_to_f16(const float* src, half_t* dst, long n, hipStream_t s) {
    int bs = 256;
    long blocks = (n + bs - 1) / bs;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)blocks), dim3(bs), 0, s, src, dst, n);
}

}  // namespace wlx
