// chain3.hip — what makes a short streaming launch slower than the 2.1 us ideal of chain.hip? Variants of the same
// 48-workgroup x 256-thread x 6-load kernel with: straight-line code bloat before / after the load issue (instruction
// fetch of cold code), a 256-byte kernarg struct, an LDS cross-wave reduce + barrier, many small side loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BigArg { const f32x4* w; long n16; float* out; const float* dep; const float* side[8]; long pad[16]; };

#define NOPS(n) asm volatile(".rept " #n "\n s_nop 0\n .endr" ::: "memory")

template <int PRE, int POST, bool LDSRED, bool SIDE>
__global__ __launch_bounds__(256) void var_kernel(BigArg a) {
    __shared__ float red[4][64];
    const int tid = threadIdx.x;
    if (PRE == 1) NOPS(256); else if (PRE == 2) NOPS(1024); else if (PRE == 3) NOPS(2048);
    const f32x4* base = a.w + (long)blockIdx.x * a.n16;
    f32x4 v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = __builtin_nontemporal_load(base + (long)i * 256 + tid);
    float d = a.dep[tid & 63];
    float sd = 0.f;
    if (SIDE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sd += a.side[i][tid & 63];
    }
    if (POST == 1) NOPS(256); else if (POST == 2) NOPS(1024); else if (POST == 3) NOPS(2048);
    float s = d + sd;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    if (LDSRED) {
        red[tid >> 6][tid & 63] = s;
        __syncthreads();
        if (tid >= 64) return;
        s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
    if (s == 12345.678f || tid < 64) a.out[blockIdx.x * 64 + (tid & 63)] = s;
}

static float time_graph(hipGraphExec_t exec, hipStream_t st, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}
static float chain_us(hipStream_t st, int nk, const std::function<void(int)>& launch) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < nk; ++k) launch(k);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    float us = time_graph(ex, st, 20);
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    return us / nk;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t pool_bytes = 640ull << 20;
    f32x4* pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0, pool_bytes));
    float* out; CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(out, 0, 1 << 20));
    float* side[8]; for (int i = 0; i < 8; ++i) { CK(hipMalloc(&side[i], 4096)); CK(hipMemset(side[i], 0, 4096)); }
    const int WGS = 48; const long n16 = 256 * 6;
    auto mk = [&](int k) { BigArg a{}; a.w = pool + (long)(k % 300) * n16 * WGS; a.n16 = n16; a.out = out + (size_t)(k & 1) * 131072;
                           a.dep = out + (size_t)((k + 1) & 1) * 131072; for (int i = 0; i < 8; ++i) a.side[i] = side[i]; return a; };
#define RUN(name, ...) printf("%-44s %6.2f us per launch\n", name, chain_us(st, 300, [&](int k) { BigArg a = mk(k); hipLaunchKernelGGL((var_kernel<__VA_ARGS__>), dim3(WGS), dim3(256), 0, st, a); }))
    RUN("base (big kernarg)", 0, 0, false, false);
    RUN("+ LDS reduce + barrier", 0, 0, true, false);
    RUN("+ 8 side loads from 8 allocations", 0, 0, false, true);
    RUN("+ both", 0, 0, true, true);
    RUN("1 KB of code before the loads", 1, 0, false, false);
    RUN("4 KB of code before the loads", 2, 0, false, false);
    RUN("8 KB of code before the loads", 3, 0, false, false);
    RUN("1 KB of code after the load issue", 0, 1, false, false);
    RUN("4 KB of code after the load issue", 0, 2, false, false);
    RUN("8 KB of code after the load issue", 0, 3, false, false);
    RUN("4 KB after + LDS reduce + side loads", 0, 2, true, true);
    // alternating two different kernels (instruction cache reuse across launches?)
    printf("%-44s %6.2f us per launch\n", "alternating two 4 KB-after variants", chain_us(st, 300, [&](int k) { BigArg a = mk(k);
        if (k & 1) hipLaunchKernelGGL((var_kernel<0, 2, true, false>), dim3(WGS), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((var_kernel<0, 2, false, true>), dim3(WGS), dim3(256), 0, st, a); }));
    return 0;
}
