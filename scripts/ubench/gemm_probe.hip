// gemm_probe.hip — where does an encoder GEMM launch spend its time? The production kernel (csrc/gemm.hip, included
// verbatim) is compiled once per variant with ONE pipeline phase removed (-DWLX_PROBE_NO_{GLOAD,LREAD,MFMA,LSTORE,BARRIER,
// EPILOGUE}); results are wrong by construction, only the time matters. Shapes: the Whisper-small encoder's layer GEMMs.
// build: for v in "" NO_GLOAD NO_LREAD NO_MFMA NO_LSTORE NO_BARRIER NO_EPILOGUE; do hipcc --offload-arch=gfx950 -O3 -std=c++17 ${v:+-DWLX_PROBE_$v} -o gemm_probe_${v:-BASE} gemm_probe.hip; done
#include "../../whisperlive_amd/csrc/gemm.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace wlx;

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int M = 1500;
    struct Shape { const char* name; int N, K, mode; } shapes[] = {
        {"fc1  N3072 K768  gelu ", 3072, 768, GEMM_GELU_F16}, {"qkv  N2304 K768  store", 2304, 768, GEMM_STORE_F16},
        {"out  N768  K768  resid", 768, 768, GEMM_RESID_F32}, {"fc2  N768  K3072 resid", 768, 3072, GEMM_RESID_F32}};
    half_t *A, *W, *C; float *X, *bias;
    CK(hipMalloc(&A, (size_t)1536 * 3072 * 2)); CK(hipMalloc(&W, (size_t)3072 * 3072 * 2)); CK(hipMalloc(&C, (size_t)1536 * 3072 * 2));
    CK(hipMalloc(&X, (size_t)1536 * 3072 * 4)); CK(hipMalloc(&bias, 3072 * 4));
    CK(hipMemsetAsync(A, 0x11, (size_t)1536 * 3072 * 2, st)); CK(hipMemsetAsync(W, 0x12, (size_t)3072 * 3072 * 2, st));
    CK(hipMemsetAsync(X, 0, (size_t)1536 * 3072 * 4, st)); CK(hipMemsetAsync(bias, 0, 3072 * 4, st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (const Shape& sh : shapes) {
        GemmParams p{};
        p.A = A; p.lda = sh.K; p.strideA = 0; p.Wp = W; p.KT = sh.K / 32; p.M = M; p.N = sh.N; p.mode = sh.mode; p.bias = bias;
        p.C = C; p.ldc = sh.N; p.X = X; p.ldx = sh.N; p.d = 768; p.qscale = 0.125f; p.rows_per_item = 1500;
        for (int i = 0; i < 5; ++i) launch_gemm(p, 1, st);
        CK(hipEventRecord(a, st));
        const int it = 50;
        for (int i = 0; i < it; ++i) launch_gemm(p, 1, st);
        CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double us = ms * 1000.0 / it, tf = 2.0 * M * sh.N * sh.K / (us * 1e-6) / 1e12;
        printf("%s %8.2f us  %7.1f TFLOP/s\n", sh.name, us, tf);
    }
    return 0;
}
