// persist_layer.hip — would a PERSISTENT decode layer (one workgroup per CU, in-launch all-to-all edges) beat the launch chain
// for a Whisper-small decoder layer? (DESIGN.md §4 argued "no" from the guide's price list only.)
//
// The dependency structure, the bytes and the workgroup counts of one Whisper-small decoder layer at 5 beam rows, t = 32, are
// reproduced phase by phase (table below); the arithmetic is a stand-in of the right size (every published value depends on
// every gathered value and on every streamed weight byte, so nothing can be elided and the two forms can be checked against
// each other), because what bounds a decode layer on MI355X is the dependent edges, not the MFMAs.
//
//   persistent : ONE launch of 256 workgroups x 256 threads (1 per CU: 96 KiB of dynamic LDS each), every workgroup walks the
//                phase table for L layers; an edge = the MI355X guide's R2 recipe: 8-byte {epoch tag, 32-bit value} granules
//                stored with relaxed agent-scope atomics (sc1, write-through), consumers re-read their granules until every
//                tag is the epoch (no flags, no fences); a workgroup's weights for its NEXT active phase are requested
//                (non-temporal) BEFORE it starts polling, so the weight stream overlaps the wait at the edge; 1 or 4 polling
//                waves per workgroup (SWEEP_WAVES).
//   launches   : the same table as one kernel launch per phase inside a hipGraph (plain loads / stores, the launch boundary is
//                the dependency), i.e. the structure the engine has today (which fuses some of these phases: 7 launches per
//                layer, 30.2 us measured, profiles/r2w_*).
// Every spin is bounded (a broken protocol reports failure instead of hanging the box).
// build: hipcc --offload-arch=gfx950 -O3 -o persist_layer persist_layer.hip ; run: ./persist_layer [layers=12] [iters=50] [fused]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Phase {
    int wg0, wg1;       // active workgroups [wg0, wg1)
    int src;            // edge buffer gathered from (-1: the layer input = buffer NBUF-1 of the previous layer)
    int gcount;         // 32-bit values gathered per active workgroup
    int gslice;         // 0: values [0, gcount) of the buffer; 1: slice (wg - wg0) * gcount (wrapping inside the buffer)
    int wq;             // weight stream: 16-byte loads per thread (x 256 threads x 16 B = bytes per workgroup)
    int dst;            // edge buffer published to
    int pcount;         // 32-bit values published per active workgroup, at (wg - wg0) * pcount
    const char* name;
};
// Whisper-small decoder layer, 5 rows (fp16 pairs = one 32-bit value; the fp32 residual stream = one value per element):
//  x: 5 x 768 fp32 = 3840 values; a 16-column fp16 tile of 5 rows = 40 values; fp32 tile = 80 values.
//  Cross attention: (head, 4 key splits) = 48 workgroups, each with 96 KiB of K/V (prefetchable: it does not depend on the
//  decoder state) -> split partials (5 x 64 fp16 + 5 x 2 fp32 = 170 values), combined per head (12 workgroups), as the
//  all-gather of 8160 partial values by each of the 48 output-projection workgroups would cost 8 sweeps.
//  MLP output projection K-split 4 ways (192 workgroups each gather a quarter of h), reduced by 48 workgroups.
enum { B_X = 0, B_QKV, B_ATT, B_X2, B_QC, B_PART, B_XO, B_X3, B_H, B_SLAB, NBUF };
static const Phase kPhases[] = {
    {48, 192, -1, 3840, 0, 6, B_QKV, 40, "LN1+QKV (144 wg, 24 KiB each)"},
    {192, 204, B_QKV, 480, 1, 1, B_ATT, 160, "self-attention (12 heads)"},
    {204, 252, B_ATT, 1920, 0, 6, B_X2, 80, "out-proj + residual (48)"},
    {0, 48, B_X2, 3840, 0, 6, B_QC, 40, "LN2 + cross query (48)"},
    {48, 96, B_QC, 160, 1, 24, B_PART, 170, "cross attention (12 heads x 4 splits, 96 KiB K/V each)"},
    {96, 108, B_PART, 680, 1, 0, B_XO, 160, "split combine (12)"},
    {144, 192, B_XO, 1920, 0, 6, B_X3, 80, "cross out-proj + residual (48)"},
    {0, 192, B_X3, 3840, 0, 6, B_H, 40, "LN3 + fc1 + GELU (192)"},
    {0, 192, B_H, 1920, 1, 6, B_SLAB, 80, "fc2, 4 K slices (192)"},
    {192, 240, B_SLAB, 320, 1, 0, B_X, 80, "slice reduce + residual (48)"},
};
// The SAME layer with the engine's fusions (7 launches per layer today): cross-attention query projection fused into the
// (head, 8 splits) attention workgroups (each re-reads its head's 96 KiB of query weights), the split combine fused into the
// output projection's prologue (every one of its 48 workgroups gathers all 96 x 170 partial values), the MLP output
// projection K-split 2 ways with the partial slabs summed by their consumers (the next layer's first projection gathers
// x + 2 slabs, the next residual update its own 3 x 80 values).
static const Phase kPhasesFused[] = {
    {48, 192, -1, 11520, 0, 6, B_QKV, 40, "LN1(x + 2 slabs) + QKV (144)"},
    {192, 204, B_QKV, 480, 1, 1, B_ATT, 160, "self-attention (12 heads)"},
    {204, 252, B_ATT, 1920, 0, 6, B_X2, 80, "out-proj + residual (48)"},
    {0, 96, B_X2, 3840, 0, 36, B_PART, 170, "LN2 + cross query (96 KiB) + cross attention (48 KiB K/V): 12 heads x 8 splits"},
    {96, 144, B_PART, 16320, 0, 6, B_X3, 80, "split combine + cross out-proj + residual (48)"},
    {0, 192, B_X3, 3840, 0, 6, B_H, 40, "LN3 + fc1 + GELU (192)"},
    {144, 240, B_H, 3840, 1, 12, B_X, 120, "fc2, 2 K slices -> slabs (96)"},
};
constexpr int NPH_MAX = sizeof(kPhases) / sizeof(kPhases[0]);
constexpr int BUF_VALUES = 16384;          // capacity of one edge buffer (values)
constexpr int MAXQ = 36;
__constant__ Phase dPhases[NPH_MAX];
__constant__ int dNPH;
static int NPH = NPH_MAX;

__device__ __forceinline__ float mix(float a, float b) { return a * 0.731f + b * 0.269f + 0.01f; }

// the stand-in arithmetic of one phase instance, shared by both forms: act (LDS, n values), weights (registers) -> pcount values
template <bool PERSIST>
__device__ __forceinline__ void phase_compute(const Phase& ph, const float* act, int n, const f32x4* wreg, float* red, int tid,
                                              float* out /* LDS, pcount */) {
    float s = 0.f;
    for (int i = tid; i < n; i += 256) s += act[i];
    float ws = 0.f;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q)
        if (q < ph.wq) ws += (wreg[q][0] + wreg[q][1]) + (wreg[q][2] + wreg[q][3]);
    s = s + ws * 1e-4f;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int j = tid; j < ph.pcount; j += 256) out[j] = mix(tot * (1.0f / 4096.0f), act[j % n]) + 1e-3f * (float)(j & 7);
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ persistent form
template <int SWEEP_WAVES>
__global__ __launch_bounds__(256) void persist_kernel(gu64* gran /* [2][NBUF][BUF_VALUES] */, const f32x4* __restrict__ weights,
                                                      long wstride16, int layers, unsigned* fail, float* result) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* act = reinterpret_cast<float*>(smem);                 // up to 16384 gathered values
    float* outv = act + 16384;                                    // up to 256 published values
    float* red = outv + 256;
    volatile int* sfail = reinterpret_cast<volatile int*>(red + 8);
    const int wg = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) sfail[0] = 0;
    __syncthreads();
    f32x4 wreg[MAXQ];
    // phases this workgroup takes part in
    int mine[NPH_MAX], nm = 0;
    const int NPH = dNPH;
    for (int p = 0; p < NPH; ++p) if (wg >= dPhases[p].wg0 && wg < dPhases[p].wg1) mine[nm++] = p;
    if (nm == 0) return;
    auto request_weights = [&](int layer, int p) {
        const Phase& ph = dPhases[p];
        const f32x4* base = weights + ((long)(layer * NPH + p) * 256 + wg) * wstride16;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) if (q < ph.wq) wreg[q] = __builtin_nontemporal_load(base + (long)q * 256 + tid);
    };
    request_weights(0, mine[0]);
    bool dead = false;
    for (int layer = 0; layer < layers && !dead; ++layer) {
        for (int k = 0; k < nm && !dead; ++k) {
            const int p = mine[k];
            const Phase ph = dPhases[p];
            // ---- gather: epoch of the producing phase instance (the layer input comes from the previous layer's last phase)
            const int sp = (ph.src < 0) ? NPH - 1 : p - 1;
            const int sl = (ph.src < 0) ? layer - 1 : layer;
            const int sbuf = (ph.src < 0) ? B_X : ph.src;
            if (sl >= 0) {
                const unsigned epoch = (unsigned)(sl * NPH + sp + 1);
                gu64* g = gran + ((long)(sl & 1) * NBUF + sbuf) * BUF_VALUES;
                const int off = ph.gslice ? ((wg - ph.wg0) * ph.gcount) : 0;
                const int cap = dPhases[sp].pcount * (dPhases[sp].wg1 - dPhases[sp].wg0);
                const int nthr = SWEEP_WAVES * 64;
                if (tid < nthr) {
                    // passes of 16 granule loads in flight per lane (the guide's sweep): a pass is re-read until all its tags match
                    for (int b0 = 0; b0 < ph.gcount && !dead; b0 += nthr * 16) {
                        unsigned spins = 0;
                        for (;;) {
                            unsigned long long x[16];
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const int i = b0 + u * nthr + tid;
                                x[u] = __hip_atomic_load(g + (off + (i < ph.gcount ? i : b0)) % cap, RLX_AGENT);
                            }
                            bool ok = true;
#pragma unroll
                            for (int u = 0; u < 16; ++u) ok = ok && ((unsigned)(x[u] >> 32) == epoch);
                            if (__all(ok)) {
#pragma unroll
                                for (int u = 0; u < 16; ++u) { const int i = b0 + u * nthr + tid; if (i < ph.gcount) act[i] = __uint_as_float((unsigned)x[u]); }
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1u << 18)) { if ((tid & 63) == 0) { *fail = 1000u + (unsigned)p; sfail[0] = 1; } dead = true; break; }
                        }
                    }
                }
            } else {
                for (int i = tid; i < ph.gcount; i += 256) act[i] = 0.25f + 1e-4f * (float)i;      // the first layer's input
            }
            __syncthreads();
            if (sfail[0]) { dead = true; break; }
            phase_compute<true>(ph, act, ph.gcount, wreg, red, tid, outv);
            // ---- the NEXT active phase's weights are requested before this phase publishes (they overlap the next wait)
            {
                const int nk = (k + 1 < nm) ? k + 1 : 0;
                const int nl = (k + 1 < nm) ? layer : layer + 1;
                if (nl < layers) request_weights(nl, mine[nk]);
            }
            // ---- publish
            const unsigned epoch = (unsigned)(layer * NPH + p + 1);
            gu64* g = gran + ((long)(layer & 1) * NBUF + ph.dst) * BUF_VALUES + (long)(wg - ph.wg0) * ph.pcount;
            for (int j = tid; j < ph.pcount; j += 256)
                __hip_atomic_store(g + j, ((unsigned long long)epoch << 32) | __float_as_uint(outv[j]), RLX_AGENT);
            if (layer == layers - 1 && p == NPH - 1 && tid < ph.pcount) result[(wg - ph.wg0) * ph.pcount + tid] = outv[tid];
        }
    }
}

// ------------------------------------------------------------------------------------------------ launch-chain form
__global__ __launch_bounds__(256) void phase_kernel(int p, int layer, const float* __restrict__ src, int cap, float* dst,
                                                    const f32x4* __restrict__ weights, long wstride16, float* result, int last) {
    extern __shared__ __attribute__((aligned(16))) char smem2[];
    float* act = reinterpret_cast<float*>(smem2);
    float* outv = act + 16384;
    float* red = outv + 256;
    const Phase ph = dPhases[p];
    const int NPH = dNPH;
    const int wg = blockIdx.x + ph.wg0, tid = threadIdx.x;
    f32x4 wreg[MAXQ];
    const f32x4* base = weights + ((long)(layer * NPH + p) * 256 + wg) * wstride16;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) if (q < ph.wq) wreg[q] = __builtin_nontemporal_load(base + (long)q * 256 + tid);
    const int off = ph.gslice ? ((wg - ph.wg0) * ph.gcount) : 0;
    if (src) {
        for (int b0 = 0; b0 < ph.gcount; b0 += 256 * 16) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int i = b0 + u * 256 + tid; x[u] = src[(off + (i < ph.gcount ? i : b0)) % cap]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int i = b0 + u * 256 + tid; if (i < ph.gcount) act[i] = x[u]; }
        }
    }
    else { for (int i = tid; i < ph.gcount; i += 256) act[i] = 0.25f + 1e-4f * (float)i; }
    __syncthreads();
    phase_compute<false>(ph, act, ph.gcount, wreg, red, tid, outv);
    for (int j = tid; j < ph.pcount; j += 256) dst[(long)(wg - ph.wg0) * ph.pcount + j] = outv[j];
    if (last && tid < ph.pcount) result[(wg - ph.wg0) * ph.pcount + tid] = outv[tid];
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 12;
    const int iters = argc > 2 ? atoi(argv[2]) : 50;
    const bool fused = argc > 3 && !strcmp(argv[3], "fused");
    const Phase* kP = fused ? kPhasesFused : kPhases;
    NPH = fused ? (int)(sizeof(kPhasesFused) / sizeof(kPhasesFused[0])) : NPH_MAX;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(dPhases), kP, sizeof(Phase) * NPH));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(dNPH), &NPH, sizeof(int)));
    printf("phase table: %s\n", fused ? "the engine's fused structure (7 phases per layer)" : "one phase per dependency edge (10 per layer)");
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // weights: a distinct region per (layer, phase, workgroup): MAXQ x 256 x 16 B = 96 KiB -> 12 layers x 10 phases x 256 wg = 2.9 GB
    const long wstride16 = (long)MAXQ * 256;
    const size_t wbytes = (size_t)layers * NPH * 256 * wstride16 * 16;
    f32x4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemsetAsync(w, 0x11, wbytes, st));
    gu64* gran; const size_t gbytes = (size_t)2 * NBUF * BUF_VALUES * 8;
    CK(hipMalloc((void**)&gran, gbytes));
    float* plain; CK(hipMalloc(&plain, (size_t)NBUF * BUF_VALUES * 4 * 2));
    unsigned* fail; CK(hipMalloc(&fail, 4));
    float *res_p, *res_l; CK(hipMalloc(&res_p, 65536)); CK(hipMalloc(&res_l, 65536));
    double layer_w = 0, layer_g = 0;
    for (int p = 0; p < NPH; ++p) {
        const int n = kP[p].wg1 - kP[p].wg0;
        layer_w += (double)n * kP[p].wq * 4096.0; layer_g += (double)n * kP[p].gcount * 8.0;
        printf("phase %d %-58s wg %3d  gathers %5d values/wg  weights %5.1f KiB/wg  publishes %3d values/wg\n", p, kP[p].name, n,
               kP[p].gcount, kP[p].wq * 4.0, kP[p].pcount);
    }
    printf("per layer: %.1f MB of weights / K,V streamed, %.2f MB of granule reads, %d in-launch edges\n", layer_w / 1e6, layer_g / 1e6, NPH);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t shm = 96 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&phase_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&persist_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&persist_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    auto run_persist = [&](int sweep, float* ms_out) -> bool {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            float tot = 0.f;
            for (int i = 0; i < iters; ++i) {
                CK(hipMemsetAsync((void*)gran, 0, gbytes, st)); CK(hipMemsetAsync(fail, 0, 4, st));
                CK(hipEventRecord(e0, st));
                if (sweep == 1) hipLaunchKernelGGL(persist_kernel<1>, dim3(256), dim3(256), shm, st, gran, w, wstride16, layers, fail, res_p);
                else hipLaunchKernelGGL(persist_kernel<4>, dim3(256), dim3(256), shm, st, gran, w, wstride16, layers, fail, res_p);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
                unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
                if (f) { printf("persistent form FAILED (bounded spin expired in phase %u)\n", f - 1000u); return false; }
            }
            best = tot / iters < best ? tot / iters : best;
        }
        *ms_out = best; return true;
    };
    // launch chain inside one graph
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < layers; ++l)
        for (int p = 0; p < NPH; ++p) {
            const Phase& ph = kP[p];
            const int sp = (ph.src < 0) ? NPH - 1 : p - 1, sl = (ph.src < 0) ? l - 1 : l, sbuf = (ph.src < 0) ? B_X : ph.src;
            const float* src = sl >= 0 ? plain + ((size_t)(sl & 1) * NBUF + sbuf) * BUF_VALUES : nullptr;
            float* dst = plain + ((size_t)(l & 1) * NBUF + ph.dst) * BUF_VALUES;
            const int cap = kP[sp].pcount * (kP[sp].wg1 - kP[sp].wg0);
            hipLaunchKernelGGL(phase_kernel, dim3(ph.wg1 - ph.wg0), dim3(256), shm, st, p, l, src, cap, dst, w, wstride16, res_l,
                               (l == layers - 1 && p == NPH - 1) ? 1 : 0);
        }
    CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
    float ms_l = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_l = ms / iters < ms_l ? ms / iters : ms_l;
    }
    printf("launch chain : %8.2f us per layer (%d launches per layer, %.2f us per launch)\n", 1e3 * ms_l / layers, NPH, 1e3 * ms_l / layers / NPH);
    const int nres = kP[NPH - 1].pcount * (kP[NPH - 1].wg1 - kP[NPH - 1].wg0);
    std::vector<float> rl(nres), rp(nres);
    CK(hipMemcpy(rl.data(), res_l, nres * 4, hipMemcpyDeviceToHost));
    for (int sweep : {1, 4}) {
        float ms_p;
        if (!run_persist(sweep, &ms_p)) continue;
        CK(hipMemcpy(rp.data(), res_p, nres * 4, hipMemcpyDeviceToHost));
        double md = 0; for (int i = 0; i < nres; ++i) md = fmax(md, fabs((double)rp[i] - rl[i]));
        printf("persistent   : %8.2f us per layer (%d polling wave%s per workgroup; %.2f us per edge)  = %.2fx of the launch chain; "
               "result max |diff| vs launch chain %.3g (%s)\n", 1e3 * ms_p / layers, sweep, sweep > 1 ? "s" : "", 1e3 * ms_p / layers / NPH,
               ms_p / ms_l, md, md < 1e-3 ? "same" : "DIFFERENT");
    }
    printf("engine today : 30.2 us per layer over 7 launches (profiles/r2w_*): adopt the persistent form only if it is <= 0.85x of that\n");
    return 0;
}
