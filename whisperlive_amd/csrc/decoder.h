// decoder.h — launchers for the autoregressive-decoder kernels (decoder.hip, search.hip).
#pragma once
#include "kernels.h"

namespace wlx {

#define WLX_MAX_MT 4          // the first-generation kernel holds up to 4 MFMA row tiles (64 rows) per launch; wider passes run as row chunks
#define WLX_MAX_DEC_ROWS 320  // decoder rows of one step (items x beams): 64 items x 5 beams (round 5: was 64 rows — a slot decoded at most 12 clips per step)
#define WLX_XSPLIT 8          // key splits of the decode cross-attention (flash-decoding)
#define WLX_MAX_CAND 32       // 2*beam candidates per row
#define WLX_MAX_HYP 32        // finished hypotheses kept per item

// Row tables (device, int32[rows]) describing what each decoder row is in this pass:
//   token   : token id fed at this row
//   pos     : its position in the text context (positional embedding, KV-cache column)
//   cache   : KV-cache row that receives this row's new K/V
//   ancrow  : row of the ancestry table used to READ the history of this row
// plus the ancestry table anc[cache_rows][448] (int16): anc[r][p] = KV-cache row that holds
// position p of the hypothesis currently living in row r (beam reordering without moving the cache).
struct RowTables {
    const int* token; const int* pos; const int* cache; const int* ancrow;
    const short* anc;
    int* intok;               // [cache_rows][448] token fed by cache row r at position p
};

void launch_dec_embed(const half_t* tok_emb, const float* pos_emb, int d, const RowTables& rt, int rows,
                      float* x, const int* done, hipStream_t s);

enum GemvIn : int { GEMV_IN_LN = 0, GEMV_IN_F16 = 1, GEMV_IN_XATTN = 2 };
enum GemvOut : int { GEMV_OUT_F16 = 0, GEMV_OUT_GELU_F16 = 1, GEMV_OUT_F32 = 2, GEMV_OUT_RESID = 3, GEMV_OUT_QKV = 4,
                     GEMV_OUT_SLAB = 5 };   // K-split partial sums, reduced by the CONSUMERS (xsrc = GEMV_X_SLABS)
// where the fp32 rows of the residual stream come from (lean kernel; GEMV_IN_LN prologue and GEMV_OUT_RESID epilogue):
//   PLAIN  the rows in X / Xres;
//   SLABS  those rows + the WLX_FC2_KS partial-sum slabs the K-split MLP output projection of the previous layer left
//          (nobody materialises the sum until the next residual update writes it back);
//   EMBED  token embedding + position (layer 0: the separate embedding launch folded into the first projection's prologue).
enum GemvXsrc : int { GEMV_X_PLAIN = 0, GEMV_X_SLABS = 1, GEMV_X_EMBED = 2 };
#ifndef WLX_CQ_GB_LDS
#define WLX_CQ_GB_LDS 1       // dec_cq_cross_attn_kernel: LayerNorm gamma / beta requested once per workgroup and shared through LDS (0 = once per wave: A/B)
#endif
#ifndef WLX_CQ_SWAP
#define WLX_CQ_SWAP 1         // dec_cq_cross_attn_kernel: softmax max / sum over the four lane rows by v_permlane swaps (0 = ds_bpermute: A/B)
#endif
#ifndef WLX_STAGE_WAVE
#define WLX_STAGE_WAVE 1      // dec_gemv2_kernel, fp16 rows in: every wave stages its own K slice of the rows, no workgroup barrier before the MFMAs (0 = cooperative copy + barrier: A/B)
#endif
#ifndef WLX_XCOMB_WAVE
#define WLX_XCOMB_WAVE 1      // dec_gemv2_kernel, cross-attention output projection: every wave combines the split partials of its own K slice, no helper waves, no barrier (0 = cooperative combine: A/B)
#endif
#ifndef WLX_FC2_KS
#define WLX_FC2_KS 2          // K slices of the lean MLP output projection (compile time: the consumers unroll over the slabs)
#endif

struct GemvParams {
    int in_mode, out_mode;
    int M;                                   // live rows (of one row chunk: <= 48 on the lean kernel)
    int Mtot;                                // (lean kernel) 0, or the rows of the whole pass, walked in row chunks, see dec_gemv2_kernel
    int chunk;                               // (set by the launcher) rows per chunk: 48 (prompt prefill, grid.z) or 16 (batched decode steps)
    int busy_device;                         // (set by the engine) three or more slots are live on this device: prefer work-saving launch shapes (gemv2_cfg)
    int rt_nz, rt_tiles, rt_magic;           // (set by the launcher) 16-row chunks folded into blockIdx.x: chunks, live n-tile workgroups, 65536 / rt_nz + 1
    int K, KT, N;                            // K real, KT = Kpad/32, N real outputs
    int KTW;                                 // (set by the launcher) k-tiles per wave, even
    int NCH;                                 // (set by the launcher, lean kernel) chunks of CH k-tiles per wave
    int nwm;                                 // (set by the launcher, lean kernel, XATTN) waves that stream weights
    int xstage;                              // (set by the launcher, lean kernel, F16) activation rows staged through LDS
    const half_t* Wp; const float* bias;
    // GEMV_IN_LN
    const float* X; long ldx; const float* gamma; const float* beta;
    // GEMV_IN_F16
    const half_t* Xh; long ldxh;
    // GEMV_IN_XATTN: split partials of the cross attention
    const half_t* part_o; const float* part_ml; int H; int R;   // [item][H][split][16][64] fp16 normalised O; [item][H][16][split][2] (m, l)
    // outputs
    half_t* Yh; long ldyh;
    float* Y; long ldy;
    float* Xres; long ldxres;
    // GEMV_OUT_QKV
    int d; float qscale; half_t* Kc; half_t* Vc; long cache_row_stride;
    const int* row_cache; const int* row_pos;
    const int* done;
    // residual-stream source (GemvXsrc) of the LN prologue / RESID epilogue, and the K-split form (GEMV_OUT_SLAB)
    int xsrc;
    float* slab; long slab_stride;           // [WLX_FC2_KS][rows][ldslab = N of the producer] fp32 partial sums; floats between slabs
    int KTS;                                 // (set by the launcher, GEMV_OUT_SLAB) k-tiles per K slice; grid.y = KT / KTS slices
    // GEMV_X_EMBED: rows gathered from the tables; workgroup 0 also writes them to X (the later residual updates read them)
    // and records the fed token (RowTables::intok)
    const half_t* tok_emb; const float* pos_emb; const int* emb_token; int* intok;   // (position / cache row: row_pos / row_cache)
    WLX_TR_FIELD
};
#ifdef WLX_TRACE
// host side of the trace: device buffer + running launch index (baked into the kernel arguments at capture time)
extern unsigned long long* g_trace_buf;
extern int g_trace_seq;
extern const char* g_trace_names[512];
static inline WlxTrace trace_next(const char* name) {
    WlxTrace t{g_trace_buf, g_trace_seq};
    if (g_trace_seq < 512) g_trace_names[g_trace_seq] = name;
    ++g_trace_seq;
    return t;
}
#define WLX_TR_ARG(name) , trace_next(name)
#else
#define WLX_TR_ARG(name)
#endif
void launch_dec_gemv(const GemvParams& p, hipStream_t s);
// kernel name (as rocprofv3 prints it) the launcher picks for these parameters — profiling hook
const char* dec_gemv_kernel_name(const GemvParams& p);
// WLX_DECODE_V1=1 selects the first-generation decode kernels (kept as the in-tree A/B reference)
extern bool g_decode_v1;
// true when launch_dec_gemv runs these parameters on the lean kernel (the only one that knows xsrc / GEMV_OUT_SLAB)
bool dec_gemv_is_lean(const GemvParams& p);
// K slices the lean kernel would cut an M x K -> N residual projection into (0: keep the single RESID launch).
// WLX_FC2_KS=0 in the environment turns the split off (A/B).
int dec_gemv_slab_split(int M, int K, int N);

// causal self-attention over the KV cache, one wave per (row, head)
// ident_ancestry: ancrow[r] == r for every row of this pass (decode steps), see dec_self_attn2_kernel
void launch_dec_self_attn(const half_t* q, long ldq, const half_t* Kc, const half_t* Vc, long cache_row_stride,
                          int d, int H, const RowTables& rt, int rows, half_t* out, long ldo,
                          const int* done, bool ident_ancestry, hipStream_t s);
// cross-attention of R rows per item against the item's 1500 encoder keys, split over keys; groups of R (<=16) rows,
// group_item[g] = audio item whose K/V group g attends to. Kp / Vp: tile-packed cross K / V of ONE decoder layer
// (gemm.hip GEMM_CROSS_KV), item_stride halfs per item.
void launch_dec_cross_attn(const half_t* q, long ldq, const half_t* Kp, const half_t* Vp, long item_stride, int H, int R,
                           int groups, int rows, const int* group_item, half_t* part_o, float* part_ml, hipStream_t s);

// fused LayerNorm + cross-attention query projection + cross-attention partials: one launch for the two above it when
// dec_cq_cross_attn_eligible (Whisper-small shapes)
bool dec_cq_cross_attn_eligible(int d, int H, int R);
void launch_dec_cq_cross_attn(const float* X, long ldx, const float* gamma, const float* beta, const half_t* Wp, const float* bias,
                              float qscale, int d, const half_t* Kp, const half_t* Vp, long item_stride, int H, int R, int groups,
                              int rows, const int* group_item, half_t* part_o, float* part_ml, hipStream_t s);
// combine of the cross attention's split partials into fp16 rows out[M][H*64] (batched rows: see decoder.hip)
void launch_dec_xattn_combine(const half_t* part_o, const float* part_ml, int M, int H, int R, half_t* out, long ldo, hipStream_t s);
// raw cross-attention scores of head h (tile-packed K of one layer AND item) for `rows` query rows -> out[rows][1536] fp32
void launch_dec_align_scores(const half_t* q, long ldq, const half_t* Kp_item, int h, int rows, float* out, hipStream_t s);

// ---------------------------------------------------------------- search.hip
struct SearchState {            // device pointers, one set per slot
    int* step;                  // [1] decode step counter
    int* done;                  // [1] all items finished
    int* done_host;             // pinned host word raised together with `done` (the host polls it; no per-step copy)
    int* n_finished;            // [1]
    int* item_done;             // [items]
    int* plen;                  // [items] prompt length (first generated position)
    float* cum;                 // [rows] cumulative log-prob of the hypothesis in each row
    int* row_done;              // [rows] (sampling mode)
    float* cand_score; int* cand_tok;  // [rows][WLX_MAX_CAND]
    int* samp_tok; float* samp_lp;     // [rows]
    int* hyp_tokens;            // [items][WLX_MAX_HYP][448]  — hyp_tokens, hyp_len, hyp_score, no_speech live in PINNED HOST memory (round 6): written by
                                // the update kernels, never read on the device; the host reads them when it sees done_host, without waiting for the stream
    int* hyp_len;               // [items][WLX_MAX_HYP]
    float* hyp_score;           // [items][WLX_MAX_HYP]  (normalised)
    int* n_hyp;                 // [items]
    int* n_hyp_host;            // [items] the same count in the pinned result area (the host reads the results there, see hyp_tokens)
    float* no_speech;           // [items]
    int* nsp_row;               // [rows] 1 if this row's raw distribution defines no_speech_prob (-1 none)
    // row tables (mutable here)
    int* token; int* pos; short* anc; int* intok;
    // chunked scan of the logits rows (beam mode, search_scan_kernel -> search_merge_update_kernel)
    float* scan_stats;          // [rows][SC_MAXCH][SC_NSTAT]
    float* scan_cv; int* scan_ci;   // [rows][SC_MAXCH + 1][WLX_MAX_CAND] per-chunk candidate lists (value desc, id asc)
    // rule state carried from step to step (third-generation search): per row (last generated token was a timestamp,
    // the one before it was — or fewer than two generated tokens —, latest generated timestamp token or -1, unused)
    int* rule;                  // [rows][4]
};
#define SC_THREADS 256
#define SC_NPT 8
#define SC_CHUNK (SC_THREADS * SC_NPT)   // 2048 vocabulary ids per scan workgroup
#define SC_MAXCH 27                      // vocab <= 53248 (+1: generation 3 cuts text and timestamp ids separately)
#define SC_NSTAT 8
struct SearchParams {
    int V; long ldl;
    int items, R, rows;
    int sampling;               // 0 beam search, 1 multinomial sampling
    int beam, ncand, max_cand_hyp /* round(beam*patience) */, num_hyp;
    int allow_early_exit;
    float length_penalty, rep_penalty, temperature;
    int no_repeat_ngram, topk;
    int suppress_blank, apply_ts_rules, max_initial_ts;
    int sot, eot, no_timestamps, ts_begin, no_speech, blank;
    int max_length;
    unsigned long long seed;
    const unsigned* suppress_mask;   // [ceil(V/32)] bit = 1 -> suppressed
};
// SearchParams live in device memory (sp_dev) so a captured step graph is reusable across calls
void launch_search_rows(const float* logits, const SearchParams* sp_dev, int rows, const SearchState& st, hipStream_t s);
void launch_search_update(const SearchParams* sp_dev, int items, const SearchState& st, hipStream_t s);
// beam mode: (chunks x rows) scan workgroups + one merge/update workgroup per item (rule state carried in SearchState::rule)
void launch_search_scan3(const float* logits, long ldl, int V, const SearchParams* sp_dev, int rows, const SearchState& st,
                         hipStream_t s);
void launch_search_merge_update3(const float* logits, long ldl, int V, const SearchParams* sp_dev, int items, int R,
                                 const SearchState& st, hipStream_t s);
// zero the per-call search state (step, done, finished counters, per-item / per-row flags, hypothesis lengths)
void launch_search_reset(const SearchState& st, int items, int rows, hipStream_t s);
// softmax over ids [0, vlim) of row r, probability of token toks[r] -> out[r]   (text_token_probs of Whisper.align)
void launch_token_prob_rows(const float* logits, long ldl, int vlim, int rows, const int* toks, float* out, hipStream_t s);
// softmax prob of token `tok` in given logits rows -> out[rows]
void launch_token_prob(const float* logits, long ldl, int V, int rows, int tok, float* out, hipStream_t s);
// softmax restricted to ids -> probs[rows][n]
void launch_lang_probs(const float* logits, long ldl, int rows, const int* ids, int n, float* probs, hipStream_t s);

}  // namespace wlx
