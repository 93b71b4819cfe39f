// chain.hip — micro-benchmarks behind the decode-step design: what does a dependent chain of short
// weight-streaming launches cost on MI355X, as a function of workgroups, bytes per launch, and next-launch prefetch?
// build: hipcc --offload-arch=gfx950 -O3 -o chain chain.hip ; run on the GPU box (scripts/ubench/run.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// each thread: NL 16-byte loads issued up front (wave-contiguous 1 KiB per load), then a dependent reduce + one store
template <int NL, bool NT>
__global__ __launch_bounds__(1024) void stream_kernel(const f32x4* __restrict__ w, long n16_per_wg, float* __restrict__ out,
                                                      const float* __restrict__ dep, const f32x4* __restrict__ pf, long pf_n16_per_wg) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const f32x4* base = w + (long)blockIdx.x * n16_per_wg;
    f32x4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        long idx = (long)i * nthr + tid;
        if (idx >= n16_per_wg) idx = n16_per_wg - 1;
        v[i] = NT ? __builtin_nontemporal_load(base + idx) : base[idx];
    }
    float d = dep[tid & 63];                // the "activation" written by the previous launch
    if (pf) {                               // prefetch the next launch's weights: loads whose results are dropped
        const f32x4* pb = pf + (long)blockIdx.x * pf_n16_per_wg;
        for (long idx = tid; idx < pf_n16_per_wg; idx += nthr) {
            f32x4 t = pb[idx];
            asm volatile("" :: "v"(t));
        }
    }
    float s = d;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    if (s == 12345.678f || tid == 0) out[blockIdx.x * 64 + (tid & 63)] = s;
}

__global__ void empty_kernel(float* out, const float* dep) { if (threadIdx.x == 0 && dep[0] == 7.f) out[0] = 1.f; }

static float time_graph(hipGraphExec_t exec, hipStream_t st, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / iters;
}

template <class F>
static float chain_us(hipStream_t st, int nk, F&& launch) {
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
    const size_t pool_bytes = 640ull << 20;      // > MALL (256 MiB): a cyclic walk never hits
    f32x4* pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0, pool_bytes));
    float* out; CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(out, 0, 1 << 20));
    const int NK = 96;
    printf("empty kernel chain: %.2f us per launch\n", chain_us(st, NK, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(48), dim3(256), 0, st, out, out + 64); }));

    struct Cfg { int wgs, thr, nl; };
    // bytes per launch = wgs * thr * nl * 16
    const Cfg cfgs[] = {
        {48, 256, 6}, {48, 512, 6}, {48, 1024, 6}, {96, 256, 6}, {96, 512, 6}, {192, 256, 6}, {192, 512, 6}, {256, 256, 6}, {256, 512, 6},
        {512, 256, 6}, {512, 512, 6}, {1024, 256, 6}, {256, 256, 2}, {256, 256, 12}, {256, 512, 12}, {512, 256, 12}, {1024, 256, 12}, {2048, 256, 12},
        {256, 1024, 12}, {4096, 256, 12}, {4096, 256, 6},
    };
    printf("%5s %5s %3s %9s | %8s %8s %8s %8s | %8s (us per launch; GB/s for plain)\n", "wgs", "thr", "nl", "KB", "plain", "nt", "pf-next", "mall", "GB/s");
    for (const Cfg& c : cfgs) {
        const long n16 = (long)c.thr * c.nl;                 // 16-byte words per workgroup
        const long per_launch = n16 * c.wgs;                 // words per launch
        const long nslots = (long)(pool_bytes / 16) / per_launch - 1;
        auto run = [&](int mode) {
            // modes 0..2 walk > 400 MB per replay (HBM, like the real 333 MB step); mode 3 re-reads < 100 MB (MALL-resident)
            long nk = (mode == 3) ? (100l << 20) / (per_launch * 16) : (420l << 20) / (per_launch * 16) + 1;
            if (nk > 420) nk = 420;
            if (nk < 8) nk = 8;
            if (nk > nslots) nk = nslots;
            return chain_us(st, (int)nk, [&](int k) {
                const f32x4* w = pool + (long)(k % nslots) * per_launch;
                const f32x4* pf = (mode == 2) ? pool + (long)((k + 1) % nslots) * per_launch : nullptr;
                float* o = out + (size_t)(k & 1) * 65536 * 2;
                const float* dep = out + (size_t)((k + 1) & 1) * 65536 * 2;
#define L(NL_, NT_) hipLaunchKernelGGL((stream_kernel<NL_, NT_>), dim3(c.wgs), dim3(c.thr), 0, st, w, n16, o, dep, pf, n16)
                if (c.nl == 2) { if (mode == 1) L(2, true); else L(2, false); }
                else if (c.nl == 6) { if (mode == 1) L(6, true); else L(6, false); }
                else { if (mode == 1) L(12, true); else L(12, false); }
#undef L
            });
        };
        // walk a fresh region each launch: nslots * per_launch covers the pool, 96 launches * up to 200 MB wraps
        const float a = run(0), b = run(1), p = run(2), m = run(3);
        printf("%5d %5d %3d %9.0f | %8.2f %8.2f %8.2f %8.2f | %8.0f\n", c.wgs, c.thr, c.nl, per_launch * 16 / 1024.0, a, b, p, m,
               per_launch * 16 / (a * 1e-6) / 1e9);
    }
    return 0;
}
