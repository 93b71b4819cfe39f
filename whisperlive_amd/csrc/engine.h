// engine.h — host-side state of libwlx: weights in kernel layout, per-slot device buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/wlx.h"
#include "decoder.h"

namespace wlx {

// writes the thread-local message wlx_last_error() returns; shared by every translation unit of the library
int set_error(int code, const char* fmt, ...);
// per-device non-blocking stream for set-up work (see engine.hip: the legacy stream is never used)
hipStream_t util_stream();

struct EncLayerW {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    half_t *Wqkv, *Wo, *W1, *W2;
    float *bqkv, *bo, *b1, *b2;
};
struct DecLayerW {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
    half_t *Wqkv, *Wo, *Wcq, *Wco, *W1, *W2;
    float *bqkv, *bo, *bcq, *bco, *b1, *b2;
};

struct StepGraphKey {
    int rows, R, groups;
    bool operator<(const StepGraphKey& o) const {
        if (rows != o.rows) return rows < o.rows;
        if (R != o.R) return R < o.R;
        return groups < o.groups;
    }
};

// decode-step profiler (wlx_debug_profile_step): pass 1 only lists the launches of a step (name, algorithmic bytes);
// pass 2 replays, per kernel name, a captured graph holding just that kernel's launches of the step
struct ProfRec { std::string name; double bytes; };
struct Prof { std::vector<ProfRec> recs; bool list_only = true; std::string only; int t = 0; };

struct Slot {
    std::mutex call_mu;      // held by the entry point currently using the slot (engine.hip: slot_acquire)
    int B = 0, R = 0, rows_cap = 0, cache_rows = 0, groups_cap = 0;
    hipStream_t stream = nullptr;
    bool dedicated_queue = false;       // the stream owns a hardware queue (engine.hip create_slot_stream); counted per device
    int device_of = -1;
    bool counted = false;               // in the per-device live-slot count
    unsigned promote_epoch = 0;         // the last "crowd left" event this slot tried to take a hardware queue at (engine.hip slot_acquire)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_lm0 = nullptr, ev_lm1 = nullptr;   // the last log-mel launch (its time is read lazily: wlx_logmel_resident does not wait)
    hipEvent_t ev_en0 = nullptr, ev_en1 = nullptr;   // the last encoder pass (wlx_encode does not wait either)
    bool en_pending = false;
    bool lm_pending = false;
    bool gen_pending = false;                  // the last generate's device time / step count: read in wlx_timings_get, not in wlx_generate
    std::vector<int> lm_items;          // items whose log-mel was requested and not launched yet (engine.hip flush_logmel)
    bool busy_variant = false;          // the decode launches of this slot use the work-saving shapes (three or more live slots on the device; engine.hip device_is_busy)
    std::vector<void*> allocs;
    // features
    float* pcm = nullptr; size_t pcm_cap = 0;       // [B][pcm_cap]
    float* feats = nullptr; long feat_ld = 0;        // [B][n_mels][feat_ld]
    std::vector<int> nframes;
    std::vector<int64_t> npcm;                       // samples resident per item
    unsigned* gmax = nullptr;
    long long* d_rng = nullptr;                      // [2 * WLX_LM_MAXRANGES] range table of the last wlx_logmel_ring
    // encoder
    half_t *featT = nullptr, *h1 = nullptr, *ln = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr,
           *attn = nullptr, *h2 = nullptr, *enc16 = nullptr;
    float *x = nullptr, *enc32 = nullptr;
    long featT_stride = 0, h1_stride = 0;
    half_t *ck = nullptr, *cvt = nullptr;            // cross K [L][B][TPAD][d], V^T [L][B][d][TPAD]
    int enc_batch = 0;
    // decoder
    half_t *kc = nullptr, *vc = nullptr;             // self cache [L][cache_rows][448][d]
    float* xd = nullptr; half_t *qd = nullptr, *attnd = nullptr, *hd = nullptr;
    float* slab = nullptr;                           // [WLX_FC2_KS][48][d] partial sums of the K-split MLP output projection (decoder.hip GEMV_OUT_SLAB)
    half_t* part_o = nullptr;
    float *part_ml = nullptr, *logits = nullptr;
    long ldl = 0;
    int *d_token = nullptr, *d_pos = nullptr, *d_cache = nullptr, *d_ancrow = nullptr, *d_group_item = nullptr;
    // the decoder pass's working set (scratch rows + row tables): `step` = the decode steps and chunked passes (rows_cap = 64
    // rows, the members above), `pf` = the one-pass prompt prefill (up to 448 rows: engine.hip prefill_tokens)
    struct DecBufs {
        float* xd; half_t *qd, *attnd, *hd; float* slab; int slab_rows; half_t* part_o; float* part_ml;
        int *d_token, *d_pos, *d_cache, *d_ancrow, *d_group_item;
    };
    DecBufs pf{};
    bool pf_one_pass = false;           // set around the one-pass prompt prefill's decoder pass (engine.hip prefill_tokens): no K-split MLP projection there
    bool pf_ok = false;                 // every projection of this model runs on the lean kernel in row chunks (decided at creation)
    short* d_anc = nullptr; int* d_intok = nullptr;
    bool anc_ident = false;                          // the uploaded row tables have ancrow[r] == r (decode steps; upload_rows)
    SearchState st{};
    SearchParams* d_sp = nullptr;
    unsigned* d_suppress = nullptr;
    int* d_lang_ids = nullptr; float* d_probs = nullptr; float* d_tokprob = nullptr;
    // pinned host staging
    int* h_stage = nullptr; size_t h_stage_ints = 0;
    // pinned staging of wlx_generate: set-up tables in (one async copy each, no synchronisation) and results out
    unsigned char* h_gen = nullptr; size_t h_gen_bytes = 0;
    int* h_pf = nullptr; bool h_pf_used = false;   // pinned staging of the one-pass prompt prefill's row tables (its own: no wait for the stream before the first prefill of a call)
    int* h_hyp = nullptr;                      // pinned result area the update kernels write: [n_hyp B | hyp_len B*H | hyp_score B*H | no_speech B | hyp_tokens B*H*448]
    int max_items = 0;                         // B of that layout
    std::vector<int> last_suppress; bool suppress_valid = false;   // the suppress mask on the device was built from this list
    std::map<StepGraphKey, hipGraphExec_t> graphs;
    wlx_timings tm{};
    Prof* prof = nullptr;
    // word alignment (wlx_align): while `align` is set, decoder_pass also writes the raw cross-attention scores of the
    // alignment heads for the rows of the current chunk
    struct AlignCapture { float* scores; const int32_t* heads; int n_heads, n_tok, row0, item; }* align = nullptr;
    float* align_scores = nullptr; size_t align_cap = 0;     // [n_heads][n_tok][1536] fp32, grown on demand
    int* d_align_tgt = nullptr; float* d_align_prob = nullptr;   // [448]
};

// Device-resident PCM ring of one client stream (include/wlx.h: wlx_ring_*; engine.hip). Samples [base, base + resident) of the
// stream live contiguously at buf[0 .. resident): a trim moves the survivors to the front (once per 30 s of audio: <= 1 MB device to
// device), so every reader sees plain contiguous memory. Writers: the socket thread (append / trim). Readers: kernels of the VAD
// object's stream and of a slot's stream; `mu` serialises the calls, `last_read` (recorded on the reader's stream behind its launch)
// is what a trim waits for before it moves data under a reader that was launched but has not run yet.
struct Ring {
    std::mutex mu;
    int device = 0;
    float* buf = nullptr; size_t cap = 0;
    int64_t base = 0, resident = 0;
    hipStream_t stream = nullptr;
    hipEvent_t last_read = nullptr; bool read_pending = false; hipStream_t last_read_stream = nullptr;
};

struct Engine {
    wlx_spec spec{};
    int device = 0;
    int H = 0;
    std::vector<void*> allocs;
    LogmelConsts lm{};
    half_t *conv1_w = nullptr, *conv2_w = nullptr;
    float *conv1_b = nullptr, *conv2_b = nullptr, *enc_pos = nullptr, *enc_ln_g = nullptr, *enc_ln_b = nullptr;
    int conv1_KT = 0;
    std::vector<EncLayerW> enc;
    half_t* Wckv = nullptr; float* bckv = nullptr;
    half_t *tok_emb16 = nullptr, *Wvocab = nullptr;
    float *dec_pos = nullptr, *dec_ln_g = nullptr, *dec_ln_b = nullptr;
    std::vector<DecLayerW> dec;
    std::mutex mu;
    std::vector<Slot*> slots;
    bool use_graph = true;
};

}  // namespace wlx

struct wlx_engine : public wlx::Engine {};
struct wlx_ring : public wlx::Ring {};
