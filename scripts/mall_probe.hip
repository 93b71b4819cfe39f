// Round 4 probe (not part of libwlx.so): does the batched decode step's cross attention — 55 MB of one layer's packed K / V per launch,
// 663 MB per step, more than the 256 MiB Infinity Cache — run faster when the layer's K / V were touched ~20 us earlier, and can that
// touch run as a parallel branch of the captured step graph under the latency-bound projections in front of the cross attention?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe scripts/mall_probe.hip && /tmp/mall_probe
// Emulation: kv_read = the cross attention's load pattern (1152 workgroups x 6 waves x 8 KiB, all requested up front);
// proj = a latency-bound projection (192 workgroups x 16 waves x 6 KiB of weights, nt); prefetch = 256 x 4 waves walking 55 MB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half_t;
typedef half_t f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(384) void kv_read(const half_t* __restrict__ kv, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wg = blockIdx.x;
    const half_t* p = kv + (wg * 6 + wave) * 4096 + lane * 8;     // 8 KiB per wave
    f16x8 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f16x8*>(p + i * 512);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a += (float)v[i][0] + (float)v[i][7];
    if (a == 12345.678f) sink[0] = a;
}
__global__ __launch_bounds__(1024) void proj(const half_t* __restrict__ w, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const half_t* p = w + ((long)(blockIdx.x % 48) * 16 + wave) * 3072 + lane * 8;   // 6 KiB per wave; the 4 row chunks share a weight tile
    f16x8 v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(p + i * 512));
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) a += (float)v[i][0] + (float)v[i][3];
    __shared__ float red[16];
    if (lane == 0) red[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < 16; ++i) s += red[i]; if (s == 12345.678f) sink[0] = s; }
}
// nwg x 4 waves walk `bytes` with `DEPTH` 1 KiB loads in flight per wave
template <int DEPTH>
__global__ __launch_bounds__(256) void prefetch(const half_t* __restrict__ kv, long bytes, float* sink) {
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * 4, wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long chunks = bytes / 1024;
    float a = 0.f;
    for (long c0 = wid * DEPTH; c0 < chunks; c0 += nwaves * DEPTH) {
        f16x8 v[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) { const long c = (c0 + i < chunks) ? c0 + i : chunks - 1; v[i] = *reinterpret_cast<const f16x8*>(kv + c * 512 + lane * 8); }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) a += (float)v[i][0];
    }
    if (a == 12345.678f) sink[0] = a;
}
__global__ void flush_read(const float4* __restrict__ b, long n4, float* sink) {
    float a = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) a += b[i].x;
    if (a == 12345.678f) sink[0] = a;
}

static float med(std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int L = 12;
    const long KV = 1152L * 6 * 8192;                 // 56.6 MB per layer
    const long WB = 48L * 16 * 6144;                  // 4.7 MB: one projection's emulated weights (x4 per layer, distinct)
    half_t *kv, *w; float *sink, *big;
    const long BIG = 1536L << 20;
    CK(hipMalloc(&kv, KV * L)); CK(hipMalloc(&w, WB * 4 * L)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&big, BIG));
    CK(hipMemset(kv, 0, KV * L)); CK(hipMemset(w, 0, WB * 4 * L)); CK(hipMemset(big, 0, BIG));
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto flush = [&] { hipLaunchKernelGGL(flush_read, dim3(2048), dim3(256), 0, s, (const float4*)big, BIG / 16, sink); };
    auto kvk = [&](int l, hipStream_t st) { hipLaunchKernelGGL(kv_read, dim3(1152), dim3(384), 0, st, kv + (long)l * KV / 2, sink); };
    auto pj = [&](int l, int j, hipStream_t st) { hipLaunchKernelGGL(proj, dim3(192), dim3(1024), 0, st, w + ((long)l * 4 + j) * WB / 2, sink); };
    auto timeit = [&](auto&& pre, auto&& body, int reps) {
        std::vector<float> t;
        for (int r = 0; r < reps; ++r) {
            pre();
            CK(hipEventRecord(e0, s)); body(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1000.f);
        }
        return med(t);
    };
    // 1. one kv_read: cold (after 1.5 GB of other reads), warm (same layer just read), warm behind four projections
    const float empty = timeit([] {}, [] {}, 50);
    const float cold = timeit([&] { flush(); pj(0, 0, s); }, [&] { kvk(0, s); }, 30);
    const float warm = timeit([&] { flush(); kvk(0, s); pj(0, 0, s); }, [&] { kvk(0, s); }, 30);
    const float warm4 = timeit([&] { flush(); kvk(0, s); for (int j = 0; j < 4; ++j) pj(0, j, s); }, [&] { kvk(0, s); }, 30);
    const float pf8 = timeit([&] { flush(); pj(0, 0, s); }, [&] { hipLaunchKernelGGL(prefetch<8>, dim3(256), dim3(256), 0, s, kv, KV, sink); }, 30);
    const float pf8_128 = timeit([&] { flush(); pj(0, 0, s); }, [&] { hipLaunchKernelGGL(prefetch<8>, dim3(128), dim3(256), 0, s, kv, KV, sink); }, 30);
    const float pfwarm = timeit([&] { flush(); hipLaunchKernelGGL(prefetch<8>, dim3(256), dim3(256), 0, s, kv, KV, sink); for (int j = 0; j < 4; ++j) pj(0, j, s); }, [&] { kvk(0, s); }, 30);
    const float chain4 = timeit([&] { flush(); pj(0, 0, s); }, [&] { for (int j = 0; j < 4; ++j) pj(1, j, s); }, 30);
    printf("event pair with nothing between: %.2f us\n", empty);
    printf("kv_read 56.6 MB: cold %.2f us (%.2f TB/s) | warm %.2f us (%.2f TB/s) | warm behind 4 projections %.2f us | warm by prefetch kernel + 4 projections %.2f us\n",
           cold, KV / cold * 1e-6, warm, KV / warm * 1e-6, warm4, pfwarm);
    printf("prefetch kernel alone (cold): 256 wg %.2f us, 128 wg %.2f us | four projections alone %.2f us\n", pf8, pf8_128, chain4);
    // 2. a 12-layer "step": per layer 4 projections + kv_read(l); variants: plain | prefetch(l) as a forked branch under the projections
    for (int variant = 0; variant < 4; ++variant) {
        const int pwg = variant == 2 ? 128 : variant == 3 ? 64 : 256;
        hipGraph_t g; hipGraphExec_t ge;
        hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l) {
            if (variant) {
                CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0));
                hipLaunchKernelGGL(prefetch<8>, dim3(pwg), dim3(256), 0, s2, kv + (long)l * KV / 2, KV, sink);
                CK(hipEventRecord(ej, s2));
            }
            for (int j = 0; j < 4; ++j) pj(l, j, s);
            if (variant) CK(hipStreamWaitEvent(s, ej, 0));
            kvk(l, s);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<float> t;
        for (int r = 0; r < 20; ++r) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1000.f);
        }
        printf("12-layer step graph (4 projections + kv_read per layer), %s: %.1f us\n",
               variant == 0 ? "plain" : variant == 1 ? "prefetch branch 256 wg" : variant == 2 ? "prefetch branch 128 wg" : "prefetch branch 64 wg", med(t));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
