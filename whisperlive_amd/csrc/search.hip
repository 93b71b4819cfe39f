// search.hip — device-side token search for ctranslate2.models.Whisper.generate
// (whisper_live/transcriber/transcriber_faster_whisper.py:1380-1407): logits processors
// (suppress_tokens, suppress_blank, repetition penalty, no-repeat-ngram, Whisper timestamp rules
// with max_initial_timestamp_index), log-softmax, beam search with patience / length penalty
// (T = 0) or multinomial sampling of num_hypotheses independent rows (T > 0), no_speech_prob.
// CTranslate2's source is not in the reference tree; the rules are restated from the published
// OpenAI definition (whisper/decoding.py ApplyTimestampRules, SuppressBlank, SuppressTokens;
// same as HF generation/logits_process.py:1909-2047) and the CT2 beam-search contract described in
// SURVEY.md Appendix A.5 — oracle/decoding.py is the bit-for-bit CPU statement of this file.
//
// Everything runs on the GPU so a decode step never synchronises with the host: search_rows (one
// workgroup per decoder row) turns a logits row into its 2*beam best continuations (or one
// sample), search_update (one workgroup per audio item) merges them, retires finished
// hypotheses, reorders the beams by rewriting the int16 ancestry table (the KV cache itself never
// moves) and publishes the next input tokens / the done flag.
#include "decoder.h"

namespace wlx {

#define SR_THREADS 1024
#define SR_NPT 52   // values per thread: vocab <= 53248

__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int w = 1; w < SR_THREADS / 64; ++w) r = fmaxf(r, scratch[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < SR_THREADS / 64; ++w) r += scratch[w];   // fixed order: deterministic
    return r;
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(SR_THREADS) void search_rows_kernel(const float* __restrict__ logits,
                                                                 const SearchParams* __restrict__ spp, SearchState st) {
    if (*st.done) return;
    const SearchParams sp = *spp;   // device-resident: the captured step graph is reusable across calls
    const int r = blockIdx.x;
    const int item = r / sp.R;
    const int rb = r - item * sp.R;
    if (st.item_done[item]) return;
    if (sp.sampling) { if (rb >= sp.num_hyp || st.row_done[r]) return; }
    else if (rb >= sp.beam) return;

    __shared__ int hist[WLX_T_TEXT];
    __shared__ unsigned seen[(SR_THREADS * SR_NPT + 31) / 32];
    __shared__ float fs[16];
    __shared__ int is_[16];
    __shared__ int last_ts_idx;
    __shared__ float chunk_sum[SR_THREADS];

    const int tid = threadIdx.x;
    const int V = sp.V;
    const int p = st.pos[r];
    const int plen = st.plen[item];
    const int ngen = p + 1 - plen;
    const short* ar = st.anc + (long)r * WLX_T_TEXT;
    const float* lrow = logits + (long)r * sp.ldl;

    if (tid == 0) last_ts_idx = -1;
    const bool need_seen = (sp.rep_penalty != 1.0f) || (sp.no_repeat_ngram > 0);
    if (need_seen) for (int i = tid; i < (V + 31) / 32; i += SR_THREADS) seen[i] = 0u;
    __syncthreads();
    for (int j = tid; j < ngen; j += SR_THREADS) {
        const int tk = st.intok[(long)ar[plen + j] * WLX_T_TEXT + plen + j];
        hist[j] = tk;
        if (tk >= sp.ts_begin) atomicMax(&last_ts_idx, j);
    }
    __syncthreads();

    // ---- load the row (strided: thread tid owns ids tid + i*1024)
    float v[SR_NPT];
#pragma unroll
    for (int i = 0; i < SR_NPT; ++i) {
        const int id = tid + i * SR_THREADS;
        v[i] = (id < V) ? lrow[id] : WLX_NEG_INF;
    }

    // ---- no_speech_prob from the RAW distribution at the sot position
    if (st.nsp_row[r] > 0) {
        float mx = WLX_NEG_INF;
#pragma unroll
        for (int i = 0; i < SR_NPT; ++i) mx = fmaxf(mx, v[i]);
        mx = block_max(mx, fs);
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < SR_NPT; ++i) sm += __expf(v[i] - mx);
        sm = block_sum(sm, fs);
        if (tid == 0) st.no_speech[item] = __expf(lrow[sp.no_speech] - mx) / sm;
        __syncthreads();
    }

    // ---- repetition penalty / no-repeat-ngram (off by default in the reference)
    if (sp.rep_penalty != 1.0f) {
        for (int j = tid; j < ngen; j += SR_THREADS) atomicOr(&seen[hist[j] >> 5], 1u << (hist[j] & 31));
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SR_NPT; ++i) {
            const int id = tid + i * SR_THREADS;
            if (id < V && (seen[id >> 5] >> (id & 31)) & 1u)
                v[i] = (v[i] < 0.f) ? v[i] * sp.rep_penalty : v[i] / sp.rep_penalty;
        }
        __syncthreads();
    }
    if (sp.no_repeat_ngram > 0 && ngen >= sp.no_repeat_ngram - 1) {
        const int n = sp.no_repeat_ngram;
        if (sp.rep_penalty != 1.0f) { for (int i = tid; i < (V + 31) / 32; i += SR_THREADS) seen[i] = 0u; __syncthreads(); }
        for (int j = tid; j + n - 1 < ngen; j += SR_THREADS) {
            bool match = true;
            for (int q = 0; q < n - 1; ++q) match = match && (hist[j + q] == hist[ngen - (n - 1) + q]);
            if (match) atomicOr(&seen[hist[j + n - 1] >> 5], 1u << (hist[j + n - 1] & 31));
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SR_NPT; ++i) {
            const int id = tid + i * SR_THREADS;
            if (id < V && (seen[id >> 5] >> (id & 31)) & 1u) v[i] = WLX_NEG_INF;
        }
    }

    // ---- static suppressions + timestamp rules
    const bool ts = sp.apply_ts_rules != 0;
    const bool last_was_ts = ts && ngen >= 1 && hist[ngen - 1] >= sp.ts_begin;
    const bool penult_was_ts = ts && (ngen < 2 || hist[ngen - 2] >= sp.ts_begin);
    int ts_last = -1;  // timestamps in [ts_begin, ts_last) are forbidden
    if (ts && last_ts_idx >= 0) {
        const int lt = hist[last_ts_idx];
        ts_last = (last_was_ts && !penult_was_ts) ? lt : lt + 1;
    }
    const bool first = (ngen == 0);
    const int last_allowed = (sp.max_initial_ts >= 0) ? sp.ts_begin + sp.max_initial_ts : 0x7fffffff;
#pragma unroll
    for (int i = 0; i < SR_NPT; ++i) {
        const int id = tid + i * SR_THREADS;
        if (id >= V) continue;
        bool kill = (sp.suppress_mask[id >> 5] >> (id & 31)) & 1u;
        if (first && sp.suppress_blank && (id == sp.blank || id == sp.eot)) kill = true;
        if (ts) {
            if (id == sp.no_timestamps) kill = true;
            if (last_was_ts) {
                if (penult_was_ts) { if (id >= sp.ts_begin) kill = true; }
                else if (id < sp.eot) kill = true;
            }
            if (id >= sp.ts_begin && id < ts_last) kill = true;
            if (first) {
                if (id < sp.ts_begin) kill = true;
                if (id > last_allowed) kill = true;
            }
        }
        if (kill) v[i] = WLX_NEG_INF;
    }

    // ---- statistics: all / text [0, ts_begin) / timestamps [ts_begin, V)
    float mx_all = WLX_NEG_INF, mx_text = WLX_NEG_INF, mx_ts = WLX_NEG_INF;
#pragma unroll
    for (int i = 0; i < SR_NPT; ++i) {
        const int id = tid + i * SR_THREADS;
        mx_all = fmaxf(mx_all, v[i]);
        if (id < sp.ts_begin) mx_text = fmaxf(mx_text, v[i]);
        else mx_ts = fmaxf(mx_ts, v[i]);
    }
    mx_all = block_max(mx_all, fs);
    float lse_sel, mx_sel;
    bool text_masked = false;
    if (ts) {
        mx_text = block_max(mx_text, fs);
        mx_ts = block_max(mx_ts, fs);
        float s_ts = 0.f;
        if (mx_ts > WLX_NEG_INF) {
#pragma unroll
            for (int i = 0; i < SR_NPT; ++i) {
                const int id = tid + i * SR_THREADS;
                if (id >= sp.ts_begin && id < V) s_ts += __expf(v[i] - mx_ts);
            }
        }
        s_ts = block_sum(s_ts, fs);
        const float lse_ts = (mx_ts > WLX_NEG_INF) ? mx_ts + __logf(s_ts) : WLX_NEG_INF;
        // "if the probability mass on timestamps exceeds every single text token, sample a timestamp"
        if (lse_ts > mx_text) {
            text_masked = true;
#pragma unroll
            for (int i = 0; i < SR_NPT; ++i) {
                const int id = tid + i * SR_THREADS;
                if (id < sp.ts_begin) v[i] = WLX_NEG_INF;
            }
        }
        if (text_masked) { lse_sel = lse_ts; mx_sel = mx_ts; }
    }
    if (!text_masked) {
        float s_all = 0.f;
#pragma unroll
        for (int i = 0; i < SR_NPT; ++i) s_all += __expf(v[i] - mx_all);
        s_all = block_sum(s_all, fs);
        lse_sel = mx_all + __logf(s_all);
        mx_sel = mx_all;
    }

    if (!sp.sampling) {
        // ---- beam search: the row's ncand best continuations, score = cum + log-prob
        const float base = st.cum[r] - lse_sel;
        float bv = WLX_NEG_INF; int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < SR_NPT; ++i) {
            const int id = tid + i * SR_THREADS;
            if (v[i] > bv) { bv = v[i]; bi = id; }   // ids ascend with i: first max = smallest id
        }
        for (int k = 0; k < sp.ncand; ++k) {
            // block argmax (value desc, id asc)
            float wv = bv; int wi = bi;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(wv, o, 64);
                const int oi = __shfl_xor(wi, o, 64);
                if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
            }
            __syncthreads();
            if ((tid & 63) == 0) { fs[tid >> 6] = wv; is_[tid >> 6] = wi; }
            __syncthreads();
            float gv = fs[0]; int gi = is_[0];
#pragma unroll
            for (int w = 1; w < SR_THREADS / 64; ++w)
                if (fs[w] > gv || (fs[w] == gv && is_[w] < gi)) { gv = fs[w]; gi = is_[w]; }
            if (tid == 0) {
                st.cand_score[(long)r * WLX_MAX_CAND + k] = (gi < V) ? gv + base : WLX_NEG_INF;
                st.cand_tok[(long)r * WLX_MAX_CAND + k] = (gi < V) ? gi : sp.eot;   // row fully masked
            }
            if (gi < V && (gi & (SR_THREADS - 1)) == tid) {
                // this thread owned the winner: retire it and rescan
                bv = WLX_NEG_INF; bi = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < SR_NPT; ++i) {
                    const int id = tid + i * SR_THREADS;
                    if (id == gi) v[i] = WLX_NEG_INF;
                    if (v[i] > bv) { bv = v[i]; bi = id; }
                }
            }
        }
    } else {
        // ---- sampling (beam_size = 1, num_hypotheses independent rows)
        int chosen = 0;
        if (sp.temperature <= 0.f || sp.topk == 1) {
            float bv = WLX_NEG_INF; int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < SR_NPT; ++i) {
                const int id = tid + i * SR_THREADS;
                if (v[i] > bv) { bv = v[i]; bi = id; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            __syncthreads();
            if ((tid & 63) == 0) { fs[tid >> 6] = bv; is_[tid >> 6] = bi; }
            __syncthreads();
            float gv = fs[0]; int gi = is_[0];
#pragma unroll
            for (int w = 1; w < SR_THREADS / 64; ++w)
                if (fs[w] > gv || (fs[w] == gv && is_[w] < gi)) { gv = fs[w]; gi = is_[w]; }
            chosen = gi;
            if (tid == 0) { st.samp_tok[r] = gi; st.samp_lp[r] = gv - lse_sel; }
        } else {
            // inverse-CDF over softmax((v - mx)/T) in natural id order: the masked row is written
            // back, re-read in contiguous chunks of SR_NPT ids per thread, chunk sums scanned by
            // thread 0 (fixed order => reproducible for a given seed).
            float* wrow = const_cast<float*>(lrow);
#pragma unroll
            for (int i = 0; i < SR_NPT; ++i) {
                const int id = tid + i * SR_THREADS;
                if (id < V) wrow[id] = v[i];
            }
            __syncthreads();
            const float invT = 1.0f / sp.temperature;
            float cs = 0.f;
            for (int q = 0; q < SR_NPT; ++q) {
                const int id = tid * SR_NPT + q;
                if (id < V) cs += __expf((wrow[id] - mx_sel) * invT);
            }
            chunk_sum[tid] = cs;
            __syncthreads();
            if (tid == 0) {
                float total = 0.f;
                for (int t2 = 0; t2 < SR_THREADS; ++t2) total += chunk_sum[t2];
                const unsigned long long h = splitmix64(sp.seed ^ splitmix64(((unsigned long long)r << 32) | (unsigned)p));
                const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
                const float target = u * total;
                float run = 0.f; int ch = 0;
                for (; ch < SR_THREADS - 1; ++ch) { if (run + chunk_sum[ch] > target) break; run += chunk_sum[ch]; }
                int pick = -1, lastvalid = -1;
                for (int q = 0; q < SR_NPT; ++q) {
                    const int id = ch * SR_NPT + q;
                    if (id >= V) break;
                    const float e = __expf((wrow[id] - mx_sel) * invT);
                    if (e > 0.f) lastvalid = id;
                    run += e;
                    if (run > target && e > 0.f) { pick = id; break; }
                }
                if (pick < 0) pick = (lastvalid >= 0) ? lastvalid : sp.eot;
                st.samp_tok[r] = pick;
                st.samp_lp[r] = wrow[pick] - lse_sel;
                is_[0] = pick;
            }
            __syncthreads();
            chosen = is_[0];
        }
        (void)chosen;
    }
}

void launch_search_rows(const float* logits, const SearchParams* sp_dev, int rows, const SearchState& st, hipStream_t s) {
    hipLaunchKernelGGL(search_rows_kernel, dim3(rows), dim3(SR_THREADS), 0, s, logits, sp_dev, st);
}

// ------------------------------------------------------------------ per-item bookkeeping
__global__ __launch_bounds__(256) void search_update_kernel(const SearchParams* __restrict__ spp, SearchState st) {
    if (*st.done) return;
    const SearchParams sp = *spp;
    const int item = blockIdx.x;
    const int tid = threadIdx.x;
    if (item == 0 && tid == 0) atomicAdd(st.step, 1);
    if (st.item_done[item]) return;
    const int r0 = item * sp.R;
    const int plen = st.plen[item];

    __shared__ short anc_s[16 * WLX_T_TEXT];     // staged ancestry rows of this item (R <= 16)
    __shared__ int parent[16], newtok[16];
    __shared__ float newcum[16];
    __shared__ int hyp_src[16], hyp_extra[16], hyp_slot[16], hyp_n;
    __shared__ int n_active_s, finished_s;

    if (!sp.sampling) {
        const int p = st.pos[r0];
        const int ngen = p + 1 - plen;
        const int max_new = sp.max_length - plen;
        const bool is_last = (ngen + 1 >= max_new);
        if (tid == 0) {
            // merge beam*ncand candidates -> best ncand (score desc; ties: lower row, lower rank)
            bool used[16 * WLX_MAX_CAND];
            for (int i = 0; i < sp.beam * sp.ncand; ++i) used[i] = false;
            int n_active = 0, nh_new = 0, n_hyp = st.n_hyp[item];
            bool top_beam_finished = false;
            for (int k = 0; k < sp.ncand; ++k) {
                float bs = WLX_NEG_INF; int bb = -1, bj = -1;
                for (int b = 0; b < sp.beam; ++b)
                    for (int j = 0; j < sp.ncand; ++j) {
                        if (used[b * sp.ncand + j]) continue;
                        const float sc = st.cand_score[(long)(r0 + b) * WLX_MAX_CAND + j];
                        if (bb < 0 || sc > bs) { bs = sc; bb = b; bj = j; }
                        break;  // each row's list is sorted: only its first unused entry can win
                    }
                if (bb < 0) break;
                used[bb * sp.ncand + bj] = true;
                const int tok = st.cand_tok[(long)(r0 + bb) * WLX_MAX_CAND + bj];
                if (tok == sp.eot || is_last) {
                    if (k >= sp.beam) continue;
                    if (n_hyp + nh_new < WLX_MAX_HYP && nh_new < 16) {
                        const int len = ngen + ((tok == sp.eot) ? 0 : 1);
                        const int slot = n_hyp + nh_new;
                        hyp_src[nh_new] = bb;
                        hyp_extra[nh_new] = (tok == sp.eot) ? -1 : tok;
                        hyp_slot[nh_new] = slot;
                        st.hyp_len[item * WLX_MAX_HYP + slot] = len;
                        const float denom = powf((float)(len > 0 ? len : 1), sp.length_penalty);
                        st.hyp_score[item * WLX_MAX_HYP + slot] = bs / denom;
                        ++nh_new;
                    }
                    if (k == 0) top_beam_finished = true;
                } else if (n_active < sp.beam) {
                    parent[n_active] = bb; newtok[n_active] = tok; newcum[n_active] = bs;
                    ++n_active;
                }
            }
            n_hyp += nh_new;
            st.n_hyp[item] = n_hyp;
            hyp_n = nh_new;
            n_active_s = n_active;
            bool fin = is_last || n_active == 0;
            if (sp.allow_early_exit) fin = fin || (top_beam_finished && n_hyp >= sp.num_hyp);
            else fin = fin || (n_hyp >= sp.max_cand_hyp);
            finished_s = fin ? 1 : 0;
        }
        __syncthreads();
        // stage the item's ancestry rows (old state) in LDS
        for (int i = tid; i < sp.beam * (p + 1); i += 256) {
            const int b = i / (p + 1), q = i - b * (p + 1);
            anc_s[b * WLX_T_TEXT + q] = st.anc[(long)(r0 + b) * WLX_T_TEXT + q];
        }
        __syncthreads();
        // write out newly finished hypotheses (history through the OLD ancestry)
        for (int hh = 0; hh < hyp_n; ++hh) {
            const int b = hyp_src[hh];
            int* dst = st.hyp_tokens + ((long)item * WLX_MAX_HYP + hyp_slot[hh]) * WLX_T_TEXT;
            for (int j = tid; j < ngen; j += 256)
                dst[j] = st.intok[(long)anc_s[b * WLX_T_TEXT + plen + j] * WLX_T_TEXT + plen + j];
            if (tid == 0 && hyp_extra[hh] >= 0) dst[ngen] = hyp_extra[hh];
        }
        if (finished_s) {
            if (tid == 0) {
                st.item_done[item] = 1;
                const int nf = atomicAdd(st.n_finished, 1) + 1;
                if (nf >= sp.items) *st.done = 1;
            }
            return;
        }
        // reorder beams: new row j inherits parent[j]'s history, then appends itself at p+1
        const int na = n_active_s;
        for (int i = tid; i < na * (p + 1); i += 256) {
            const int j = i / (p + 1), q = i - j * (p + 1);
            st.anc[(long)(r0 + j) * WLX_T_TEXT + q] = anc_s[parent[j] * WLX_T_TEXT + q];
        }
        if (tid < sp.beam) {
            const int j = tid;
            if (j < na) {
                st.anc[(long)(r0 + j) * WLX_T_TEXT + p + 1] = (short)(r0 + j);
                st.token[r0 + j] = newtok[j];
                st.cum[r0 + j] = newcum[j];
            } else {
                st.anc[(long)(r0 + j) * WLX_T_TEXT + p + 1] = (short)(r0 + j);
                st.token[r0 + j] = sp.eot;
                st.cum[r0 + j] = WLX_NEG_INF;
            }
            st.pos[r0 + j] = p + 1;
            st.nsp_row[r0 + j] = 0;
        }
    } else {
        // sampling: rows are independent hypotheses; slot = row index within the item
        __shared__ int alive;
        if (tid == 0) alive = 0;
        __syncthreads();
        for (int b = 0; b < sp.num_hyp; ++b) {
            const int r = r0 + b;
            if (st.row_done[r]) continue;           // uniform across the block
            const int p = st.pos[r];
            const int ngen = p + 1 - plen;
            const bool is_last = (ngen + 1 >= sp.max_length - plen);
            const int tok = st.samp_tok[r];
            const float cum = st.cum[r] + st.samp_lp[r];
            const bool fin = (tok == sp.eot) || is_last;
            if (fin) {
                int* dst = st.hyp_tokens + ((long)item * WLX_MAX_HYP + b) * WLX_T_TEXT;
                const short* ar = st.anc + (long)r * WLX_T_TEXT;
                for (int j = tid; j < ngen; j += 256) dst[j] = st.intok[(long)ar[plen + j] * WLX_T_TEXT + plen + j];
                if (tid == 0) {
                    const int len = ngen + ((tok == sp.eot) ? 0 : 1);
                    if (tok != sp.eot) dst[ngen] = tok;
                    st.hyp_len[item * WLX_MAX_HYP + b] = len;
                    st.hyp_score[item * WLX_MAX_HYP + b] = cum / powf((float)(len > 0 ? len : 1), sp.length_penalty);
                    st.row_done[r] = 1;
                    st.cum[r] = cum;
                }
            } else if (tid == 0) {
                st.cum[r] = cum;
                st.token[r] = tok;
                st.pos[r] = p + 1;
                st.anc[(long)r * WLX_T_TEXT + p + 1] = (short)r;
                st.nsp_row[r] = 0;
                alive = 1;
            }
            __syncthreads();
        }
        __syncthreads();
        if (tid == 0 && !alive) {
            st.n_hyp[item] = sp.num_hyp;
            st.item_done[item] = 1;
            const int nf = atomicAdd(st.n_finished, 1) + 1;
            if (nf >= sp.items) *st.done = 1;
        }
    }
}

void launch_search_update(const SearchParams* sp_dev, int items, const SearchState& st, hipStream_t s) {
    hipLaunchKernelGGL(search_update_kernel, dim3(items), dim3(256), 0, s, sp_dev, st);
}


// ================================================================== second generation (beam mode)
// search_rows_kernel gives one workgroup per decoder row the whole 52 K-entry logits row: five CUs read
// 207 KB each and then run ~25 block-wide reductions back to back (~100 us per step on MI355X). Here the row is
// cut into SC_CHUNK-id chunks, one 256-thread workgroup per (chunk, row) — 130 workgroups for beam 5 — which apply
// the SAME logits processors, and reduce their chunk to
//   stats : (max, sum exp) of the text ids and of the timestamp ids, plus the raw pair for no_speech_prob,
//   lists : the chunk's ncand best (value desc, id asc) allowed ids — and, for the one chunk that straddles
//           timestamp_begin, a second list restricted to timestamp ids (used when the timestamp rule masks text).
// search_merge_update_kernel (one workgroup per audio item) merges the chunk results of each beam row with
// wave-level reductions only (one list per lane, head-pointer merge), which reproduces search_rows' candidate list,
// and then runs search_update's bookkeeping unchanged from LDS. Sums are re-associated per chunk (fp32, fixed order:
// deterministic), every comparison and tie-break is the same.

__device__ __forceinline__ void sc_argmax_wave(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// block top-k of w[] (ids id0 + i*SC_THREADS + tid, ascending in i), entries with id < lo are not eligible
__device__ __forceinline__ void sc_topk(float (&w)[SC_NPT], int id0, int lo, int ncand, float* outv, int* outi,
                                        float (*sv)[4], int (*si)[4]) {
    const int tid = threadIdx.x;
    float bv = WLX_NEG_INF; int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < SC_NPT; ++i) {
        const int id = id0 + i * SC_THREADS + tid;
        if (id < lo) w[i] = WLX_NEG_INF;
        if (w[i] > bv) { bv = w[i]; bi = id; }
    }
    for (int k = 0; k < ncand; ++k) {
        float wv = bv; int wi = bi;
        sc_argmax_wave(wv, wi);
        if ((tid & 63) == 0) { sv[k & 1][tid >> 6] = wv; si[k & 1][tid >> 6] = wi; }
        __syncthreads();
        float gv = sv[k & 1][0]; int gi = si[k & 1][0];
#pragma unroll
        for (int q = 1; q < SC_THREADS / 64; ++q) {
            const float ov = sv[k & 1][q]; const int oi = si[k & 1][q];
            if (ov > gv || (ov == gv && oi < gi)) { gv = ov; gi = oi; }
        }
        if (tid == 0) { outv[k] = gv; outi[k] = gi; }
        if (gi != 0x7fffffff && ((gi - id0) & (SC_THREADS - 1)) == tid) {
            bv = WLX_NEG_INF; bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < SC_NPT; ++i) {
                const int id = id0 + i * SC_THREADS + tid;
                if (id == gi) w[i] = WLX_NEG_INF;
                if (w[i] > bv) { bv = w[i]; bi = id; }
            }
        }
    }
}

__global__ __launch_bounds__(SC_THREADS) void search_scan_kernel(const float* __restrict__ logits, long ldl, int V,
                                                                 const SearchParams* __restrict__ spp, SearchState st WLX_TR_PARAM) {
    const int chunk = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
    const int id0 = chunk * SC_CHUNK;
    WLX_TR_BEGIN();
    // the row slice first: its loads are in flight while the rule state is derived
    const float* lrow = logits + (long)r * ldl;
    float v[SC_NPT];
#pragma unroll
    for (int i = 0; i < SC_NPT; ++i) {
        const int id = id0 + i * SC_THREADS + tid;
        v[i] = (id < V) ? lrow[id] : WLX_NEG_INF;
    }
    // (no done test: the scan only writes its own scratch; search_merge_update_kernel holds the gate)
    const SearchParams sp = *spp;
    const int item = r / sp.R;
    const int rb = r - item * sp.R;
    if (st.item_done[item] || rb >= sp.beam) return;

    __shared__ int hist[WLX_T_TEXT];
    __shared__ int last_ts_idx;
    __shared__ float fs[3][4];
    __shared__ float sv[2][4];
    __shared__ int si[2][4];

    const int p = st.pos[r];
    const int plen = st.plen[item];
    const int ngen = p + 1 - plen;
    const short* ar = st.anc + (long)r * WLX_T_TEXT;
    if (tid == 0) last_ts_idx = -1;
    __syncthreads();
    for (int j = tid; j < ngen; j += SC_THREADS) {
        const int tk = st.intok[(long)ar[plen + j] * WLX_T_TEXT + plen + j];
        hist[j] = tk;
        if (tk >= sp.ts_begin) atomicMax(&last_ts_idx, j);
    }
    __syncthreads();
    WLX_TR_MARK(1);
    const int wv_ = tid >> 6;

    // ---- raw (max, sum exp) of the chunk for no_speech_prob at the sot position
    float raw_m = WLX_NEG_INF, raw_s = 0.f;
    if (st.nsp_row[r] > 0) {
        float mx = WLX_NEG_INF;
#pragma unroll
        for (int i = 0; i < SC_NPT; ++i) mx = fmaxf(mx, v[i]);
        mx = wave_max(mx);
        if ((tid & 63) == 0) fs[0][wv_] = mx;
        __syncthreads();
        raw_m = fmaxf(fmaxf(fs[0][0], fs[0][1]), fmaxf(fs[0][2], fs[0][3]));
        float sm = 0.f;
        if (raw_m > WLX_NEG_INF) {
#pragma unroll
            for (int i = 0; i < SC_NPT; ++i) sm += __expf(v[i] - raw_m);
        }
        sm = wave_sum(sm);
        if ((tid & 63) == 0) fs[1][wv_] = sm;
        __syncthreads();
        raw_s = fs[1][0] + fs[1][1] + fs[1][2] + fs[1][3];
        __syncthreads();
    }

    // ---- repetition penalty / no-repeat-ngram (off by default in the reference): direct history scans
    if (sp.rep_penalty != 1.0f) {
#pragma unroll
        for (int i = 0; i < SC_NPT; ++i) {
            const int id = id0 + i * SC_THREADS + tid;
            bool hit = false;
            for (int j = 0; j < ngen; ++j) hit = hit || (hist[j] == id);
            if (hit && id < V) v[i] = (v[i] < 0.f) ? v[i] * sp.rep_penalty : v[i] / sp.rep_penalty;
        }
    }
    if (sp.no_repeat_ngram > 0 && ngen >= sp.no_repeat_ngram - 1) {
        const int n = sp.no_repeat_ngram;
        for (int j = 0; j + n - 1 < ngen; ++j) {
            bool match = true;
            for (int q = 0; q < n - 1; ++q) match = match && (hist[j + q] == hist[ngen - (n - 1) + q]);
            if (match) {
                const int banned = hist[j + n - 1];
#pragma unroll
                for (int i = 0; i < SC_NPT; ++i)
                    if (id0 + i * SC_THREADS + tid == banned) v[i] = WLX_NEG_INF;
            }
        }
    }

    // ---- static suppressions + timestamp rules (identical to search_rows_kernel)
    const bool ts = sp.apply_ts_rules != 0;
    const bool last_was_ts = ts && ngen >= 1 && hist[ngen - 1] >= sp.ts_begin;
    const bool penult_was_ts = ts && (ngen < 2 || hist[ngen - 2] >= sp.ts_begin);
    int ts_last = -1;  // timestamps in [ts_begin, ts_last) are forbidden
    if (ts && last_ts_idx >= 0) {
        const int lt = hist[last_ts_idx];
        ts_last = (last_was_ts && !penult_was_ts) ? lt : lt + 1;
    }
    const bool first = (ngen == 0);
    const int last_allowed = (sp.max_initial_ts >= 0) ? sp.ts_begin + sp.max_initial_ts : 0x7fffffff;
#pragma unroll
    for (int i = 0; i < SC_NPT; ++i) {
        const int id = id0 + i * SC_THREADS + tid;
        if (id >= V) continue;
        bool kill = (sp.suppress_mask[id >> 5] >> (id & 31)) & 1u;
        if (first && sp.suppress_blank && (id == sp.blank || id == sp.eot)) kill = true;
        if (ts) {
            if (id == sp.no_timestamps) kill = true;
            if (last_was_ts) {
                if (penult_was_ts) { if (id >= sp.ts_begin) kill = true; }
                else if (id < sp.eot) kill = true;
            }
            if (id >= sp.ts_begin && id < ts_last) kill = true;
            if (first) {
                if (id < sp.ts_begin) kill = true;
                if (id > last_allowed) kill = true;
            }
        }
        if (kill) v[i] = WLX_NEG_INF;
    }

    // ---- chunk statistics: text ids [0, ts_begin) and timestamp ids [ts_begin, V)
    float mx_text = WLX_NEG_INF, mx_ts = WLX_NEG_INF;
#pragma unroll
    for (int i = 0; i < SC_NPT; ++i) {
        const int id = id0 + i * SC_THREADS + tid;
        if (id < sp.ts_begin) mx_text = fmaxf(mx_text, v[i]);
        else mx_ts = fmaxf(mx_ts, v[i]);
    }
    mx_text = wave_max(mx_text);
    mx_ts = wave_max(mx_ts);
    if ((tid & 63) == 0) { fs[0][wv_] = mx_text; fs[1][wv_] = mx_ts; }
    __syncthreads();
    mx_text = fmaxf(fmaxf(fs[0][0], fs[0][1]), fmaxf(fs[0][2], fs[0][3]));
    mx_ts = fmaxf(fmaxf(fs[1][0], fs[1][1]), fmaxf(fs[1][2], fs[1][3]));
    __syncthreads();
    float s_text = 0.f, s_ts = 0.f;
#pragma unroll
    for (int i = 0; i < SC_NPT; ++i) {
        const int id = id0 + i * SC_THREADS + tid;
        if (id < sp.ts_begin) { if (mx_text > WLX_NEG_INF) s_text += __expf(v[i] - mx_text); }
        else if (id < V) { if (mx_ts > WLX_NEG_INF) s_ts += __expf(v[i] - mx_ts); }
    }
    s_text = wave_sum(s_text);
    s_ts = wave_sum(s_ts);
    if ((tid & 63) == 0) { fs[0][wv_] = s_text; fs[1][wv_] = s_ts; }
    __syncthreads();
    s_text = fs[0][0] + fs[0][1] + fs[0][2] + fs[0][3];
    s_ts = fs[1][0] + fs[1][1] + fs[1][2] + fs[1][3];
    if (tid == 0) {
        float* so = st.scan_stats + ((long)r * SC_MAXCH + chunk) * SC_NSTAT;
        so[0] = mx_text; so[1] = s_text; so[2] = mx_ts; so[3] = s_ts; so[4] = raw_m; so[5] = raw_s;
    }

    WLX_TR_MARK(2);
    // ---- candidate lists
    const int hi = (id0 + SC_CHUNK < V) ? id0 + SC_CHUNK : V;
    const bool mixed = ts && id0 < sp.ts_begin && sp.ts_begin < hi;
    float w[SC_NPT];
#pragma unroll
    for (int i = 0; i < SC_NPT; ++i) w[i] = v[i];
    sc_topk(w, id0, 0, sp.ncand, st.scan_cv + ((long)r * (SC_MAXCH + 1) + chunk) * WLX_MAX_CAND,
            st.scan_ci + ((long)r * (SC_MAXCH + 1) + chunk) * WLX_MAX_CAND, sv, si);
    WLX_TR_MARK(3);
    if (mixed) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SC_NPT; ++i) w[i] = v[i];
        sc_topk(w, id0, sp.ts_begin, sp.ncand, st.scan_cv + ((long)r * (SC_MAXCH + 1) + SC_MAXCH) * WLX_MAX_CAND,
                st.scan_ci + ((long)r * (SC_MAXCH + 1) + SC_MAXCH) * WLX_MAX_CAND, sv, si);
    }
    WLX_TR_END(trc);
}

void launch_search_scan(const float* logits, long ldl, int V, const SearchParams* sp_dev, int rows, const SearchState& st,
                        hipStream_t s) {
    const int nch = (V + SC_CHUNK - 1) / SC_CHUNK;
    hipLaunchKernelGGL(search_scan_kernel, dim3(nch, rows), dim3(SC_THREADS), 0, s, logits, ldl, V, sp_dev, st WLX_TR_ARG("search_scan"));
}

__global__ __launch_bounds__(256) void search_merge_update_kernel(const float* __restrict__ logits, long ldl, int V,
                                                                  const SearchParams* __restrict__ spp, SearchState st WLX_TR_PARAM) {
    if (*st.done) return;
    WLX_TR_BEGIN();
    const SearchParams sp = *spp;
    const int item = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (item == 0 && tid == 0) atomicAdd(st.step, 1);
    if (st.item_done[item]) return;
    const int r0 = item * sp.R;
    const int plen = st.plen[item];
    const int nch = (V + SC_CHUNK - 1) / SC_CHUNK;

    __shared__ float cand_s[16][WLX_MAX_CAND];
    __shared__ int cand_t[16][WLX_MAX_CAND];
    __shared__ float lv[4][SC_MAXCH + 1][WLX_MAX_CAND];
    __shared__ int li[4][SC_MAXCH + 1][WLX_MAX_CAND];
    __shared__ short anc_s[16 * WLX_T_TEXT];     // staged ancestry rows of this item (R <= 16)
    __shared__ int parent[16], newtok[16];
    __shared__ float newcum[16];
    __shared__ int hyp_src[16], hyp_extra[16], hyp_slot[16], hyp_n;
    __shared__ int n_active_s, finished_s;

    // ---------------- phase 1: per beam row, merge the chunk results into the row's ncand best continuations
    const bool ts = sp.apply_ts_rules != 0;
    const int cmix = (ts && sp.ts_begin < V && (sp.ts_begin % SC_CHUNK) != 0) ? sp.ts_begin / SC_CHUNK : -1;
    for (int rb = wave; rb < sp.beam; rb += 4) {
        const int r = r0 + rb;
        float mt = WLX_NEG_INF, st_ = 0.f, mts = WLX_NEG_INF, sts = 0.f, rm = WLX_NEG_INF, rs = 0.f;
        if (lane < nch) {
            const float* so = st.scan_stats + ((long)r * SC_MAXCH + lane) * SC_NSTAT;
            mt = so[0]; st_ = so[1]; mts = so[2]; sts = so[3]; rm = so[4]; rs = so[5];
        }
        // this lane's candidate list -> LDS (requested before the reductions below)
        if (lane <= nch) {
            const int src = (lane < nch) ? lane : SC_MAXCH;
            const float* cv = st.scan_cv + ((long)r * (SC_MAXCH + 1) + src) * WLX_MAX_CAND;
            const int* ci = st.scan_ci + ((long)r * (SC_MAXCH + 1) + src) * WLX_MAX_CAND;
            const bool have = (lane < nch) || (cmix >= 0);
            for (int k = 0; k < sp.ncand; ++k) {
                lv[wave][lane][k] = have ? cv[k] : WLX_NEG_INF;
                li[wave][lane][k] = have ? ci[k] : 0x7fffffff;
            }
        }
        const float Mtext = wave_max(mt), Mts = wave_max(mts);
        if (st.nsp_row[r] > 0) {
            const float RM = wave_max(rm);
            const float RS = wave_sum((rm > WLX_NEG_INF) ? rs * __expf(rm - RM) : 0.f);
            if (lane == 0) st.no_speech[item] = __expf(logits[(long)r * ldl + sp.no_speech] - RM) / RS;
        }
        const float S_ts = wave_sum((mts > WLX_NEG_INF) ? sts * __expf(mts - Mts) : 0.f);
        const float lse_ts = (Mts > WLX_NEG_INF) ? Mts + __logf(S_ts) : WLX_NEG_INF;
        const bool text_masked = ts && (lse_ts > Mtext);
        float lse_sel;
        if (text_masked) {
            lse_sel = lse_ts;
        } else {
            const float mx_all = fmaxf(Mtext, Mts);
            const float a = ((mt > WLX_NEG_INF) ? st_ * __expf(mt - mx_all) : 0.f) +
                            ((mts > WLX_NEG_INF) ? sts * __expf(mts - mx_all) : 0.f);
            lse_sel = mx_all + __logf(wave_sum(a));
        }
        // which lists may contribute: all ids -> the nch "all" lists; timestamps only -> pure-timestamp chunks and
        // the straddling chunk's timestamp list (slot nch)
        bool usable = false;
        if (lane < nch) {
            if (!text_masked) usable = true;
            else usable = ((long)lane * SC_CHUNK >= sp.ts_begin);
        } else if (lane == nch) {
            usable = text_masked && cmix >= 0;
        }
        const float base = st.cum[r] - lse_sel;
        int hp = 0;
        float hv = usable ? lv[wave][lane][0] : WLX_NEG_INF;
        int hi = usable ? li[wave][lane][0] : 0x7fffffff;
        for (int k = 0; k < sp.ncand; ++k) {
            float gv = hv; int gi = hi;
            sc_argmax_wave(gv, gi);
            if (lane == 0) {
                cand_s[rb][k] = (gi < V) ? gv + base : WLX_NEG_INF;
                cand_t[rb][k] = (gi < V) ? gi : sp.eot;   // row fully masked
            }
            if (usable && gi != 0x7fffffff && hi == gi) {
                ++hp;
                hv = (hp < sp.ncand) ? lv[wave][lane][hp] : WLX_NEG_INF;
                hi = (hp < sp.ncand) ? li[wave][lane][hp] : 0x7fffffff;
            }
        }
    }
    __syncthreads();
    WLX_TR_MARK(1);

    // ---------------- phase 2: search_update_kernel's beam bookkeeping, candidates read from LDS
    const int p = st.pos[r0];
    const int ngen = p + 1 - plen;
    const int max_new = sp.max_length - plen;
    const bool is_last = (ngen + 1 >= max_new);
    if (tid == 0) {
        bool used[16 * WLX_MAX_CAND];
        for (int i = 0; i < sp.beam * sp.ncand; ++i) used[i] = false;
        int n_active = 0, nh_new = 0, n_hyp = st.n_hyp[item];
        bool top_beam_finished = false;
        for (int k = 0; k < sp.ncand; ++k) {
            float bs = WLX_NEG_INF; int bb = -1, bj = -1;
            for (int b = 0; b < sp.beam; ++b)
                for (int j = 0; j < sp.ncand; ++j) {
                    if (used[b * sp.ncand + j]) continue;
                    const float sc = cand_s[b][j];
                    if (bb < 0 || sc > bs) { bs = sc; bb = b; bj = j; }
                    break;  // each row's list is sorted: only its first unused entry can win
                }
            if (bb < 0) break;
            used[bb * sp.ncand + bj] = true;
            const int tok = cand_t[bb][bj];
            if (tok == sp.eot || is_last) {
                if (k >= sp.beam) continue;
                if (n_hyp + nh_new < WLX_MAX_HYP && nh_new < 16) {
                    const int len = ngen + ((tok == sp.eot) ? 0 : 1);
                    const int slot = n_hyp + nh_new;
                    hyp_src[nh_new] = bb;
                    hyp_extra[nh_new] = (tok == sp.eot) ? -1 : tok;
                    hyp_slot[nh_new] = slot;
                    st.hyp_len[item * WLX_MAX_HYP + slot] = len;
                    const float denom = powf((float)(len > 0 ? len : 1), sp.length_penalty);
                    st.hyp_score[item * WLX_MAX_HYP + slot] = bs / denom;
                    ++nh_new;
                }
                if (k == 0) top_beam_finished = true;
            } else if (n_active < sp.beam) {
                parent[n_active] = bb; newtok[n_active] = tok; newcum[n_active] = bs;
                ++n_active;
            }
        }
        n_hyp += nh_new;
        st.n_hyp[item] = n_hyp;
        hyp_n = nh_new;
        n_active_s = n_active;
        bool fin = is_last || n_active == 0;
        if (sp.allow_early_exit) fin = fin || (top_beam_finished && n_hyp >= sp.num_hyp);
        else fin = fin || (n_hyp >= sp.max_cand_hyp);
        finished_s = fin ? 1 : 0;
    }
    // stage the item's ancestry rows (old state) in LDS while thread 0 merges
    for (int i = tid; i < sp.beam * (p + 1); i += 256) {
        const int b = i / (p + 1), q = i - b * (p + 1);
        anc_s[b * WLX_T_TEXT + q] = st.anc[(long)(r0 + b) * WLX_T_TEXT + q];
    }
    __syncthreads();
    WLX_TR_MARK(2);
    for (int hh = 0; hh < hyp_n; ++hh) {
        const int b = hyp_src[hh];
        int* dst = st.hyp_tokens + ((long)item * WLX_MAX_HYP + hyp_slot[hh]) * WLX_T_TEXT;
        for (int j = tid; j < ngen; j += 256)
            dst[j] = st.intok[(long)anc_s[b * WLX_T_TEXT + plen + j] * WLX_T_TEXT + plen + j];
        if (tid == 0 && hyp_extra[hh] >= 0) dst[ngen] = hyp_extra[hh];
    }
    if (finished_s) {
        if (tid == 0) {
            st.item_done[item] = 1;
            const int nf = atomicAdd(st.n_finished, 1) + 1;
            if (nf >= sp.items) *st.done = 1;
        }
        return;
    }
    const int na = n_active_s;
    for (int i = tid; i < na * (p + 1); i += 256) {
        const int j = i / (p + 1), q = i - j * (p + 1);
        st.anc[(long)(r0 + j) * WLX_T_TEXT + q] = anc_s[parent[j] * WLX_T_TEXT + q];
    }
    if (tid < sp.beam) {
        const int j = tid;
        st.anc[(long)(r0 + j) * WLX_T_TEXT + p + 1] = (short)(r0 + j);
        if (j < na) { st.token[r0 + j] = newtok[j]; st.cum[r0 + j] = newcum[j]; }
        else { st.token[r0 + j] = sp.eot; st.cum[r0 + j] = WLX_NEG_INF; }
        st.pos[r0 + j] = p + 1;
        st.nsp_row[r0 + j] = 0;
    }
    WLX_TR_END(trc);
}

void launch_search_merge_update(const float* logits, long ldl, int V, const SearchParams* sp_dev, int items,
                                const SearchState& st, hipStream_t s) {
    hipLaunchKernelGGL(search_merge_update_kernel, dim3(items), dim3(256), 0, s, logits, ldl, V, sp_dev, st WLX_TR_ARG("search_merge_update"));
}

// ------------------------------------------------------------------ small softmax helpers
__global__ __launch_bounds__(SR_THREADS) void token_prob_kernel(const float* __restrict__ logits, long ldl, int V,
                                                               int tok, float* __restrict__ out) {
    __shared__ float fs[16];
    const float* lrow = logits + (long)blockIdx.x * ldl;
    float mx = WLX_NEG_INF;
    for (int i = threadIdx.x; i < V; i += SR_THREADS) mx = fmaxf(mx, lrow[i]);
    mx = block_max(mx, fs);
    float sm = 0.f;
    for (int i = threadIdx.x; i < V; i += SR_THREADS) sm += __expf(lrow[i] - mx);
    sm = block_sum(sm, fs);
    if (threadIdx.x == 0) out[blockIdx.x] = __expf(lrow[tok] - mx) / sm;
}
void launch_token_prob(const float* logits, long ldl, int V, int rows, int tok, float* out, hipStream_t s) {
    hipLaunchKernelGGL(token_prob_kernel, dim3(rows), dim3(SR_THREADS), 0, s, logits, ldl, V, tok, out);
}

__global__ __launch_bounds__(256) void lang_probs_kernel(const float* __restrict__ logits, long ldl,
                                                         const int* __restrict__ ids, int n, float* __restrict__ probs) {
    __shared__ float red[4];
    const float* lrow = logits + (long)blockIdx.x * ldl;
    float mx = WLX_NEG_INF;
    for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, lrow[ids[i]]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sm = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) sm += __expf(lrow[ids[i]] - mx);
    sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
    __syncthreads();
    sm = red[0] + red[1] + red[2] + red[3];
    for (int i = threadIdx.x; i < n; i += 256) probs[(long)blockIdx.x * n + i] = __expf(lrow[ids[i]] - mx) / sm;
}
void launch_lang_probs(const float* logits, long ldl, int rows, const int* ids, int n, float* probs, hipStream_t s) {
    hipLaunchKernelGGL(lang_probs_kernel, dim3(rows), dim3(256), 0, s, logits, ldl, ids, n, probs);
}

}  // namespace wlx
