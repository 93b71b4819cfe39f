// chain2.hip — the REAL decode kernels (decoder.hip is compiled into this binary) in synthetic dependent chains:
// per-launch cost of each kernel type alone vs in the order a decoder layer runs them. Weight regions advance per
// launch through a 640 MB pool so nothing is cache-resident. build: see scripts/ubench/run.sh
#include "../../whisperlive_amd/csrc/decoder.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
using namespace wlx;

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
    const int d = 768, F = 3072, H = 12, rows = 5;
    const size_t pool_halfs = (640ull << 20) / 2;
    half_t* pool; CK(hipMalloc(&pool, pool_halfs * 2)); CK(hipMemset(pool, 0, pool_halfs * 2));
    auto dz = [&](size_t bytes) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); return p; };
    float* bias = (float*)dz(4 * 4096); float* gamma = (float*)dz(4 * 4096); float* beta = (float*)dz(4 * 4096);
    float* xd = (float*)dz(4 * 64 * d); half_t* qd = (half_t*)dz(2 * 64 * d); half_t* attnd = (half_t*)dz(2 * 64 * d);
    half_t* hd = (half_t*)dz(2 * 64 * F);
    half_t* part_o = (half_t*)dz(2ull * H * WLX_XSPLIT * 16 * 64 * 4); float* part_ml = (float*)dz(4ull * H * WLX_XSPLIT * 16 * 2 * 4);
    const long crs = (long)WLX_T_TEXT * d;
    half_t* kc = (half_t*)dz(2ull * 16 * crs); half_t* vc = (half_t*)dz(2ull * 16 * crs);
    int* d_cache = (int*)dz(64 * 4); int* d_pos = (int*)dz(64 * 4); int* d_ancrow = (int*)dz(64 * 4); int* d_tok = (int*)dz(64 * 4);
    short* d_anc = (short*)dz(2ull * 16 * WLX_T_TEXT); int* d_intok = (int*)dz(4ull * 16 * WLX_T_TEXT); int* d_gi = (int*)dz(64);
    { std::vector<int> pos(64, 32), ca(64); for (int i = 0; i < 64; ++i) ca[i] = i % 5;
      CK(hipMemcpy(d_pos, pos.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(d_cache, ca.data(), 256, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_ancrow, ca.data(), 256, hipMemcpyHostToDevice)); }
    half_t* ck = (half_t*)dz(2ull * WLX_T_AUDIO_PAD * d * 12); half_t* cvt = (half_t*)dz(2ull * WLX_T_AUDIO_PAD * d * 12);
    float* logits = (float*)dz(4ull * 16 * 53248);
    RowTables rt{d_tok, d_pos, d_cache, d_ancrow, d_anc, d_intok};

    size_t woff = 0;
    auto wnext = [&](size_t halfs) { if (woff + halfs > pool_halfs) woff = 0; half_t* p = pool + woff; woff += (halfs + 511) / 512 * 512; return p; };
    auto g_qkv = [&]() { GemvParams p{}; p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_QKV; p.M = rows; p.K = d; p.KT = d / 32; p.N = 3 * d;
        p.Wp = wnext((size_t)3 * d * d); p.bias = bias; p.X = xd; p.ldx = d; p.gamma = gamma; p.beta = beta; p.Yh = qd; p.ldyh = d; p.d = d; p.qscale = 0.125f;
        p.Kc = kc; p.Vc = vc; p.cache_row_stride = crs; p.row_cache = d_cache; p.row_pos = d_pos; launch_dec_gemv(p, st); };
    auto g_out = [&]() { GemvParams p{}; p.in_mode = GEMV_IN_F16; p.out_mode = GEMV_OUT_RESID; p.M = rows; p.K = d; p.KT = d / 32; p.N = d;
        p.Wp = wnext((size_t)d * d); p.bias = bias; p.Xh = attnd; p.ldxh = d; p.Xres = xd; p.ldxres = d; p.qscale = 1.f; launch_dec_gemv(p, st); };
    auto g_cq = [&]() { GemvParams p{}; p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_F16; p.M = rows; p.K = d; p.KT = d / 32; p.N = d;
        p.Wp = wnext((size_t)d * d); p.bias = bias; p.X = xd; p.ldx = d; p.gamma = gamma; p.beta = beta; p.Yh = qd; p.ldyh = d; p.qscale = 0.125f; launch_dec_gemv(p, st); };
    auto g_co = [&]() { GemvParams p{}; p.in_mode = GEMV_IN_XATTN; p.out_mode = GEMV_OUT_RESID; p.M = rows; p.K = d; p.KT = d / 32; p.N = d;
        p.Wp = wnext((size_t)d * d); p.bias = bias; p.part_o = part_o; p.part_ml = part_ml; p.H = H; p.R = rows; p.Xres = xd; p.ldxres = d; p.qscale = 1.f; launch_dec_gemv(p, st); };
    auto g_fc1 = [&]() { GemvParams p{}; p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_GELU_F16; p.M = rows; p.K = d; p.KT = d / 32; p.N = F;
        p.Wp = wnext((size_t)d * F); p.bias = bias; p.X = xd; p.ldx = d; p.gamma = gamma; p.beta = beta; p.Yh = hd; p.ldyh = F; p.qscale = 1.f; launch_dec_gemv(p, st); };
    auto g_fc2 = [&]() { GemvParams p{}; p.in_mode = GEMV_IN_F16; p.out_mode = GEMV_OUT_RESID; p.M = rows; p.K = F; p.KT = F / 32; p.N = d;
        p.Wp = wnext((size_t)d * F); p.bias = bias; p.Xh = hd; p.ldxh = F; p.Xres = xd; p.ldxres = d; p.qscale = 1.f; launch_dec_gemv(p, st); };
    auto g_voc = [&]() { GemvParams p{}; p.in_mode = GEMV_IN_LN; p.out_mode = GEMV_OUT_F32; p.M = rows; p.K = d; p.KT = d / 32; p.N = 51864;
        p.Wp = wnext((size_t)d * 51872); p.bias = nullptr; p.X = xd; p.ldx = d; p.gamma = gamma; p.beta = beta; p.Y = logits; p.ldy = 53248; p.qscale = 1.f; launch_dec_gemv(p, st); };
    int lay = 0;
    auto g_sa = [&]() { launch_dec_self_attn(qd, d, kc, vc, crs, d, H, rt, rows, attnd, d, nullptr, st); };
    auto g_ca = [&]() { lay = (lay + 1) % 12; launch_dec_cross_attn(qd, d, ck + (size_t)lay * WLX_T_AUDIO_PAD * d, cvt + (size_t)lay * d * WLX_T_AUDIO_PAD,
                                                                     (long)WLX_T_AUDIO_PAD * d, H, rows, 1, rows, d_gi, part_o, part_ml, st); };
    struct T { const char* name; std::function<void()> f; };
    std::vector<T> singles = {{"qkv  LN->QKV  N2304 K768 ", g_qkv}, {"self_attn2 (t=33)        ", g_sa}, {"out  F16->RES N768 K768  ", g_out},
                              {"cq   LN->F16  N768 K768  ", g_cq}, {"cross_attn               ", g_ca}, {"co   XATT->RES N768 K768 ", g_co},
                              {"fc1  LN->GELU N3072 K768 ", g_fc1}, {"fc2  F16->RES N768 K3072 ", g_fc2}, {"vocab LN->F32 N51864     ", g_voc}};
    float sum = 0.f;
    for (auto& t : singles) {
        const float us = chain_us(st, 96, [&](int) { t.f(); });
        printf("%s alone: %6.2f us per launch\n", t.name, us);
        if (&t != &singles.back()) sum += us;
    }
    printf("sum of the 8 layer kernels alone: %.2f us\n", sum);
    const float lay_us = chain_us(st, 12, [&](int) { g_qkv(); g_sa(); g_out(); g_cq(); g_ca(); g_co(); g_fc1(); g_fc2(); });
    printf("layer sequence (8 launches) x 12: %.2f us per layer\n", lay_us);
    const float gem_us = chain_us(st, 12, [&](int) { g_qkv(); g_out(); g_cq(); g_co(); g_fc1(); g_fc2(); });
    printf("layer sequence without the two attention kernels: %.2f us per layer\n", gem_us);
    const float alt_us = chain_us(st, 48, [&](int) { g_out(); g_cq(); });
    printf("out,cq alternating: %.2f us per pair\n", alt_us);
    return 0;
}
