// Which dec_gemv2_kernel instantiation the launcher picks per projection (host only, no GPU): the rules of decoder.hip gemv2_cfg made visible.
//   hipcc -std=c++17 -O1 -DWLX_AB scripts/gemv_pick_probe.cpp -o /tmp/gemv_pick_probe -Lwhisperlive_amd -l:libwlx_ab.so -Wl,-rpath,$PWD/whisperlive_amd
//   WLX_G2_CHMAX=6 /tmp/gemv_pick_probe   (libwlx_ab.so reads the A/B switches; build it with scripts/build_all.sh libwlx_ab.so:WLX_AB)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../whisperlive_amd/csrc/common.h"
#include "../whisperlive_amd/csrc/decoder.h"
using namespace wlx;
int main() {
    static float bias = 0.f;
    struct { const char* what; int in, out, M, K, N, xsrc, KS; } cases[] = {
        {"small o-proj M5", GEMV_IN_F16, GEMV_OUT_RESID, 5, 768, 768, GEMV_X_PLAIN, 0},
        {"small fc2 slab M5", GEMV_IN_F16, GEMV_OUT_SLAB, 5, 3072, 768, GEMV_X_PLAIN, WLX_FC2_KS},
        {"small o-proj slabs M5", GEMV_IN_F16, GEMV_OUT_RESID, 5, 768, 768, GEMV_X_SLABS, 0},
        {"small o-proj M60", GEMV_IN_F16, GEMV_OUT_RESID, 60, 768, 768, GEMV_X_PLAIN, 0},
        {"small fc2 slab M60", GEMV_IN_F16, GEMV_OUT_SLAB, 60, 3072, 768, GEMV_X_PLAIN, WLX_FC2_KS},
        {"small fc2 resid M60", GEMV_IN_F16, GEMV_OUT_RESID, 60, 3072, 768, GEMV_X_PLAIN, 0},
        {"small xattn M5", GEMV_IN_XATTN, GEMV_OUT_RESID, 5, 768, 768, GEMV_X_PLAIN, 0},
        {"large o-proj M5", GEMV_IN_F16, GEMV_OUT_RESID, 5, 1280, 1280, GEMV_X_PLAIN, 0},
        {"large fc2 slab M5", GEMV_IN_F16, GEMV_OUT_SLAB, 5, 5120, 1280, GEMV_X_PLAIN, WLX_FC2_KS},
        {"large fc2 resid M5", GEMV_IN_F16, GEMV_OUT_RESID, 5, 5120, 1280, GEMV_X_PLAIN, 0},
        {"large o-proj slabs M5", GEMV_IN_F16, GEMV_OUT_RESID, 5, 1280, 1280, GEMV_X_SLABS, 0},
        {"large xattn M5", GEMV_IN_XATTN, GEMV_OUT_RESID, 5, 1280, 1280, GEMV_X_PLAIN, 0},
        {"medium o-proj M5", GEMV_IN_F16, GEMV_OUT_RESID, 5, 1024, 1024, GEMV_X_PLAIN, 0},
        {"medium fc2 slab M5", GEMV_IN_F16, GEMV_OUT_SLAB, 5, 4096, 1024, GEMV_X_PLAIN, WLX_FC2_KS},
        {"medium fc2 resid M16", GEMV_IN_F16, GEMV_OUT_RESID, 16, 4096, 1024, GEMV_X_PLAIN, 0},
        {"base o-proj M5", GEMV_IN_F16, GEMV_OUT_RESID, 5, 512, 512, GEMV_X_PLAIN, 0},
        {"small mlp-up M5", GEMV_IN_LN, GEMV_OUT_GELU_F16, 5, 768, 3072, GEMV_X_PLAIN, 0},
        {"small mlp-up M16", GEMV_IN_LN, GEMV_OUT_GELU_F16, 16, 768, 3072, GEMV_X_PLAIN, 0},
        {"small mlp-up M60", GEMV_IN_LN, GEMV_OUT_GELU_F16, 60, 768, 3072, GEMV_X_PLAIN, 0},
        {"small q-proj M5", GEMV_IN_LN, GEMV_OUT_F16, 5, 768, 768, GEMV_X_PLAIN, 0},
        {"large mlp-up M5", GEMV_IN_LN, GEMV_OUT_GELU_F16, 5, 1280, 5120, GEMV_X_PLAIN, 0},
        {"large q-proj M5", GEMV_IN_LN, GEMV_OUT_F16, 5, 1280, 1280, GEMV_X_PLAIN, 0},
        {"large mlp-up M16", GEMV_IN_LN, GEMV_OUT_GELU_F16, 16, 1280, 5120, GEMV_X_PLAIN, 0},
        {"medium mlp-up M5", GEMV_IN_LN, GEMV_OUT_GELU_F16, 5, 1024, 4096, GEMV_X_PLAIN, 0},
        {"medium q-proj M5", GEMV_IN_LN, GEMV_OUT_F16, 5, 1024, 1024, GEMV_X_PLAIN, 0},
        {"small qkv slabs M5", GEMV_IN_LN, GEMV_OUT_QKV, 5, 768, 2304, GEMV_X_SLABS, 0},
        {"small qkv embed M5", GEMV_IN_LN, GEMV_OUT_QKV, 5, 768, 2304, GEMV_X_EMBED, 0},
        {"medium qkv slabs M5", GEMV_IN_LN, GEMV_OUT_QKV, 5, 1024, 3072, GEMV_X_SLABS, 0},
        {"large qkv slabs M5", GEMV_IN_LN, GEMV_OUT_QKV, 5, 1280, 3840, GEMV_X_SLABS, 0},
        {"large qkv slabs M16", GEMV_IN_LN, GEMV_OUT_QKV, 16, 1280, 3840, GEMV_X_SLABS, 0},
        {"tiny o-proj M5", GEMV_IN_F16, GEMV_OUT_RESID, 5, 384, 384, GEMV_X_PLAIN, 0},
    };
    for (auto& c : cases) {
        GemvParams p; memset(&p, 0, sizeof p);
        p.in_mode = c.in; p.out_mode = c.out; p.M = c.M; p.K = c.K; p.KT = c.K / 32; p.N = c.N; p.xsrc = c.xsrc; p.bias = &bias;
        p.H = c.K / 64; p.R = 5;
        if (c.KS) p.KTS = p.KT / c.KS;
        printf("%-24s slab_split=%d lean=%d  %s\n", c.what, dec_gemv_slab_split(c.M, c.K, c.N), (int)dec_gemv_is_lean(p), dec_gemv_kernel_name(p));
    }
    return 0;
}
