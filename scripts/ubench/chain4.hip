// chain4.hip — can the 1.6 us boundary between dependent launches be bought back? ("what is left")
//
// A decode step is ~88 launches, each of which streams weights that do NOT depend on its predecessor and then consumes
// a small activation that does. Here the same chain runs three ways inside one hipGraph:
//   serial : one stream, ordinary dependent launches (the production structure)
//   flag   : launches alternate between two captured streams (two independent chains in the graph: k, k+2, ... and
//            k+1, k+3, ...), the real dependency k -> k+1 is carried by a device-side counter: every workgroup of launch
//            k, after its output store, does an agent-scope release + atomic add; every workgroup of launch k+1 issues
//            its weight loads FIRST, then one lane spins on the counter (acquire), then the block reads the activation.
//            At most two launches are resident (each stream serialises its own), so the spin cannot deadlock as long as
//            both launches' workgroups fit on the chip at once (48..192 workgroups of 256 threads: they do).
//   noflag : the two-stream graph without the counters (WRONG results, upper bound on what overlap could give)
// Output: us per launch for each, plus a checksum check of `flag` against `serial`.
// build: hipcc --offload-arch=gfx950 -O3 -o chain4 chain4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NL = 6;            // 16-byte loads per lane (6 x 1 KiB per wave, like an N = 768 projection)

// out[k][wg*64 + lane] = sum(weights of this workgroup) + act[k-1][lane]; act[k] := out[k] of workgroup 0
template <int MODE>              // 0 serial, 1 flag, 2 noflag
__global__ __launch_bounds__(256) void link_kernel(const f32x4* __restrict__ w, long n16_per_wg, const float* act_in, float* act_out,
                                                   unsigned* flag_in, unsigned* flag_out, unsigned need) {
    const int tid = threadIdx.x;
    const f32x4* base = w + (long)blockIdx.x * n16_per_wg;
    f32x4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = __builtin_nontemporal_load(base + (long)i * 256 + tid);   // producer-independent
    if (MODE == 1) {
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(flag_in, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 18)) break;                     // bounded: a broken protocol must not hang the box
            }
        }
        __syncthreads();
    }
    float d;
    if (MODE == 1) d = __hip_atomic_load(act_in + (tid & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bypass stale L1 lines
    else d = act_in[tid & 63];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    s = s * 1e-3f + d * 0.5f + 1.0f;
    if (blockIdx.x == 0 && tid < 64) {
        if (MODE == 1) __hip_atomic_store(act_out + tid, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else act_out[tid] = s;
    }
    if (MODE == 1) {
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(flag_out, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static float time_graph(hipGraphExec_t exec, hipStream_t st, int iters, unsigned* flags, size_t flag_bytes, bool reset) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float total = 0.f;
    for (int i = 0; i < iters + 2; ++i) {
        if (reset) CK(hipMemsetAsync(flags, 0, flag_bytes, st));
        CK(hipEventRecord(a, st));
        CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 2) total += ms;
    }
    return total * 1000.f / iters;
}

int main() {
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const int NK = 88;
    const size_t pool_bytes = 640ull << 20;
    f32x4* pool; CK(hipMalloc(&pool, pool_bytes));
    {
        std::vector<float> h(pool_bytes / 4);
        unsigned x = 12345u;
        for (auto& f : h) { x = x * 1664525u + 1013904223u; f = (float)((x >> 9) & 1023) * (1.0f / 1024.0f); }
        CK(hipMemcpy(pool, h.data(), pool_bytes, hipMemcpyHostToDevice));
    }
    float* act; CK(hipMalloc(&act, (NK + 1) * 64 * sizeof(float)));
    unsigned* flags; CK(hipMalloc(&flags, (NK + 1) * 64 * sizeof(unsigned)));       // one counter per 256 B
    std::vector<float> zero((NK + 1) * 64, 0.f);
    printf("%5s %9s | %8s %8s %8s | result\n", "wgs", "KB/launch", "serial", "flag", "noflag");
    for (int wgs : {48, 96, 192}) {
        const long n16 = 256L * NL;                                       // 16-byte words per workgroup
        const long per_launch = n16 * wgs;
        auto wptr = [&](int k) { return pool + ((long)k * per_launch) % (long)(pool_bytes / 16 - per_launch); };
        float us[3]; std::vector<float> res[3];
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemcpy(act, zero.data(), zero.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemset(flags, 0, (NK + 1) * 64 * 4));
            CK(hipDeviceSynchronize());
            hipGraph_t g; hipGraphExec_t ex;
            hipEvent_t fork, join; CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
            if (mode != 0) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
            for (int k = 0; k < NK; ++k) {
                hipStream_t st = (mode == 0 || (k & 1) == 0) ? s0 : s1;
                const float* ain = act + (long)k * 64; float* aout = act + (long)(k + 1) * 64;
                unsigned* fin = flags + (long)k * 64; unsigned* fout = flags + (long)(k + 1) * 64;
                const unsigned need = (k == 0) ? 0u : (unsigned)wgs;
                if (mode == 0) hipLaunchKernelGGL((link_kernel<0>), dim3(wgs), dim3(256), 0, st, wptr(k), n16, ain, aout, fin, fout, need);
                else if (mode == 1) hipLaunchKernelGGL((link_kernel<1>), dim3(wgs), dim3(256), 0, st, wptr(k), n16, ain, aout, fin, fout, need);
                else hipLaunchKernelGGL((link_kernel<2>), dim3(wgs), dim3(256), 0, st, wptr(k), n16, ain, aout, fin, fout, need);
            }
            if (mode != 0) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
            CK(hipStreamEndCapture(s0, &g));
            CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            us[mode] = time_graph(ex, s0, 10, flags, (NK + 1) * 64 * 4, mode == 1) / NK;
            res[mode].resize(64);
            CK(hipMemcpy(res[mode].data(), act + (long)NK * 64, 64 * 4, hipMemcpyDeviceToHost));
            CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
        }
        bool same = true;
        for (int i = 0; i < 64; ++i) same = same && (res[0][i] == res[1][i]);
        printf("%5d %9.0f | %8.2f %8.2f %8.2f | flag %s serial (%.6f vs %.6f)\n", wgs, per_launch * 16 / 1024.0, us[0], us[1], us[2],
               same ? "==" : "!=", res[1][0], res[0][0]);
    }
    return 0;
}
