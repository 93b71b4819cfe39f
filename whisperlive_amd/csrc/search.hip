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
// How far the stream got, without an event (round 5): the update kernel of decode step k (item 0's workgroup, as it ends) stores k to the
// pinned word next to the done flag — a RELEASE store at system scope, behind the same thread's done-flag stores — and the host throttles
// its run-ahead on that word (engine.hip generate_impl). A hipEventRecord after every step graph + hipEventSynchronize one step behind
// cost 0.31 ms per 64-step window (27.59 vs 27.28 ms without any, profiles/r5p_*).
__device__ __forceinline__ void step_mirror(const SearchState& st, int step_no) {
    __hip_atomic_store(st.done_host + 1, step_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// An item is finished (one thread, AFTER a workgroup barrier that follows the workgroup's last result store). The results (hypothesis tokens, lengths,
// scores, no_speech_prob, counts) are in pinned host memory and the host reads them the moment it sees the done word, while the step it had already
// enqueued is still running (round 6: the result copies used to queue behind that step on the slot stream, ~0.4 ms per call that ends on an
// end-of-text). So every result store of every item must happen-before the done word: barrier (this workgroup's stores -> this thread), system-scope
// fence, the count; the thread that counts the last item fences again (the other items' workgroups fenced before they counted) and releases.
__device__ __forceinline__ void finish_item(const SearchState& st, int item, int items) {
    st.item_done[item] = 1;
    __threadfence_system();
    const int nf = atomicAdd(st.n_finished, 1) + 1;
    if (nf >= items) {
        __threadfence_system();
        *st.done = 1;
        __hip_atomic_store(st.done_host, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void search_update_kernel(const SearchParams* __restrict__ spp, SearchState st) {
    if (*st.done) return;
    const SearchParams sp = *spp;
    const int item = blockIdx.x;
    const int tid = threadIdx.x;
    int step_no = 0;                             // (item 0, thread 0) this launch's number, 1-based: mirrored to pinned memory when the workgroup ends
    if (item == 0 && tid == 0) step_no = atomicAdd(st.step, 1) + 1;
    if (st.item_done[item]) { if (step_no) step_mirror(st, step_no); return; }
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
            st.n_hyp[item] = n_hyp; st.n_hyp_host[item] = n_hyp;
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
            __syncthreads();                             // every thread's hypothesis tokens are stored
            if (tid == 0) {
                if (step_no) step_mirror(st, step_no);   // (as search_merge_update3_kernel: every update kernel stores its step number as it ends;
                finish_item(st, item, sp.items);         //  BEFORE the done word: the host may start its next call the moment it sees that one)
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
            st.n_hyp[item] = sp.num_hyp; st.n_hyp_host[item] = sp.num_hyp;
            finish_item(st, item, sp.items);
        }
    }
    if (step_no) step_mirror(st, step_no);
}

void launch_search_update(const SearchParams* sp_dev, int items, const SearchState& st, hipStream_t s) {
    hipLaunchKernelGGL(search_update_kernel, dim3(items), dim3(256), 0, s, sp_dev, st);
}


// ================================================================== beam mode: chunked scan + merge/update
// search_rows_kernel gives one workgroup per decoder row the whole 52 K-entry logits row (~100 us per step on MI355X).
// Beam mode instead cuts the row into SC_CHUNK-id chunks, one 256-thread workgroup per (chunk, row), which apply the SAME
// logits processors and reduce their chunk to (max, sum exp) statistics and its ncand best allowed ids;
// search_merge_update3_kernel (one workgroup per audio item) merges the chunk results per beam row and runs
// search_update_kernel's bookkeeping. An earlier form of this pair cost 24 + 37 us per step
// (profiles/r01e_decode_step_trace.txt): a three-level dependent load chain to rebuild the rule state from the token
// history, ~6 block-wide reductions with a barrier each, ncand block-wide argmax rounds with a barrier each, 20 scalar
// dependent loads per lane to fetch a candidate list, and a single-thread merge over a scratch array. Here
//   * the rule state of a row (was the last / the one-before-last generated token a timestamp; the latest timestamp
//     token) is carried in SearchState::rule and advanced by the merge kernel from the parent's state — no history walk
//     (repetition penalty / no-repeat-ngram, off by default in the reference, still walk it);
//   * reductions are DPP wave reductions; a chunk's statistics cost ONE barrier, its candidate list one more
//     (per-wave top-ncand with no barrier, then wave 0 merges the four lists);
//   * the merge kernel fetches lists with vector loads issued up front and merges the beams with a 16-lane argmax.
// Candidate order, tie-breaks (value desc, id asc; beam merge: score desc, row asc, list position asc) and every rule
// are those of search_rows_kernel / search_update_kernel; sums are re-associated per chunk exactly as in generation 2.
__device__ __forceinline__ float s3_wave_sum(float v) { return dpp_wave_sum(v); }
__device__ __forceinline__ float s3_wave_max(float v) { return dpp_wave_max(v); }
// (value desc, id asc) argmax over the wave, wave-uniform result: the maximum VALUE first (6 v_max_f32_dpp), then the
// smallest id among the lanes that hold it (6 v_min_i32_dpp) — branch-free; a compare-and-select on (value, id) pairs
// compiles to exec-mask control flow, ~20 instructions per butterfly step
__device__ __forceinline__ void s3_wave_argmax(float& v, int& i) {
    const float m = dpp_wave_max(v);
    // the lanes that hold the maximum: almost always exactly one -> read its id directly; only a tie (or an exhausted
    // wave: every lane at -inf) needs the second reduction for the smallest id
    const unsigned long long hit = __ballot(v == m);
    int id;
    if (__builtin_popcountll(hit) == 1) id = __builtin_amdgcn_readlane(i, (int)__builtin_ctzll(hit));
    else id = dpp_wave_min_i32((v == m) ? i : 0x7fffffff);
    i = id;
    v = m;
}
// the same over each 16-lane row separately (result in every lane of the row)
__device__ __forceinline__ void s3_row_argmax(float& v, int& i) {
    float m = v;
    WLX_DPP_REDUCE_ROW("v_max_f32_dpp", m);
    int t = (v == m) ? i : 0x7fffffff;
    WLX_DPP_REDUCE_ROW("v_min_i32_dpp", t);
    v = m; i = t;
}

// Chunk layout of generation 3: text ids [0, ts_begin) are cut into SC_CHUNK-id chunks from 0, timestamp ids
// [ts_begin, V) into chunks from ts_begin, so NO chunk straddles timestamp_begin (generation 2's straddling chunk ran
// the candidate search twice and was the critical path of the whole scan). SC_MAXCH chunks are launched; the surplus exits.
__device__ __forceinline__ void s3_chunk(int chunk, int ts_begin, int V, int& id0, int& lim, int& nct, int& nch) {
    nct = (ts_begin + SC_CHUNK - 1) / SC_CHUNK;
    nch = nct + (V - ts_begin + SC_CHUNK - 1) / SC_CHUNK;
    const bool is_ts = chunk >= nct;
    id0 = is_ts ? ts_begin + (chunk - nct) * SC_CHUNK : chunk * SC_CHUNK;
    lim = is_ts ? V : ts_begin;
}

__global__ __launch_bounds__(SC_THREADS) void search_scan3_kernel(const float* __restrict__ logits, long ldl, int V,
                                                                  const SearchParams* __restrict__ spp, SearchState st WLX_TR_PARAM) {
    const int chunk = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WLX_TR_BEGIN();
    const SearchParams sp = *spp;
    int id0, lim, nct, nch_;
    s3_chunk(chunk, sp.ts_begin, V, id0, lim, nct, nch_);
    if (chunk >= nch_) return;
    const float* lrow = logits + (long)r * ldl + id0 + tid;
    float v[SC_NPT];
#pragma unroll
    for (int i = 0; i < SC_NPT; ++i) v[i] = (id0 + i * SC_THREADS + tid < lim) ? lrow[i * SC_THREADS] : WLX_NEG_INF;
    const int item = r / sp.R;
    const int rb = r - item * sp.R;
    const int4 rule = *reinterpret_cast<const int4*>(st.rule + 4 * r);
    const int p = st.pos[r], plen = st.plen[item], nsp = st.nsp_row[r];
    if (st.item_done[item] || rb >= sp.beam) return;
    const int ngen = p + 1 - plen;

    __shared__ float fs[6][4];
    __shared__ float wv_s[4][WLX_MAX_CAND];
    __shared__ int wi_s[4][WLX_MAX_CAND];

    // ---- repetition penalty / no-repeat-ngram (off by default in the reference): history walk, as generation 2
    if (sp.rep_penalty != 1.0f || sp.no_repeat_ngram > 0) {
        __shared__ int hist[WLX_T_TEXT];
        const short* ar = st.anc + (long)r * WLX_T_TEXT;
        for (int j = tid; j < ngen; j += SC_THREADS) hist[j] = st.intok[(long)ar[plen + j] * WLX_T_TEXT + plen + j];
        __syncthreads();
        if (sp.rep_penalty != 1.0f) {
#pragma unroll
            for (int i = 0; i < SC_NPT; ++i) {
                const int id = id0 + i * SC_THREADS + tid;
                bool hit = false;
                for (int j = 0; j < ngen; ++j) hit = hit || (hist[j] == id);
                if (hit && id < lim) v[i] = (v[i] < 0.f) ? v[i] * sp.rep_penalty : v[i] / sp.rep_penalty;
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
    }
    WLX_TR_MARK(1);

    // ---- raw (max, sum exp) for no_speech_prob, BEFORE any rule
    float raw_m = WLX_NEG_INF, raw_s = 0.f;
    if (nsp > 0) {
        float mx = v[0];
#pragma unroll
        for (int i = 1; i < SC_NPT; ++i) mx = fmaxf(mx, v[i]);
        raw_m = mx;                       // per-thread max; reduced together with the rule statistics below
    }
    const float raw_local = raw_m;

    // ---- static suppressions + timestamp rules from the carried rule state
    const bool ts = sp.apply_ts_rules != 0;
    const bool last_was_ts = ts && rule.x != 0;
    const bool penult_was_ts = ts && rule.y != 0;
    int ts_last = -1;                     // timestamps in [ts_begin, ts_last) are forbidden
    if (ts && rule.z >= 0) ts_last = (last_was_ts && !penult_was_ts) ? rule.z : rule.z + 1;
    const bool first = (ngen == 0);
    const int last_allowed = (sp.max_initial_ts >= 0) ? sp.ts_begin + sp.max_initial_ts : 0x7fffffff;
    float w[SC_NPT];
#pragma unroll
    for (int i = 0; i < SC_NPT; ++i) {
        const int id = id0 + i * SC_THREADS + tid;
        bool kill = id >= lim;
        if (!kill) kill = (sp.suppress_mask[id >> 5] >> (id & 31)) & 1u;
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
        w[i] = kill ? WLX_NEG_INF : v[i];
    }

    // ---- chunk statistics (a chunk is all text or all timestamps), one barrier: maximum first (wave DPP), then the sum
    const bool ts_chunk = chunk >= nct;
    float mx = w[0];
#pragma unroll
    for (int i = 1; i < SC_NPT; ++i) mx = fmaxf(mx, w[i]);
    mx = s3_wave_max(mx);
    const float raw_wm = (nsp > 0) ? s3_wave_max(raw_local) : WLX_NEG_INF;
    if (lane == 0) { fs[0][wave] = mx; fs[2][wave] = raw_wm; }
    __syncthreads();
    mx = fmaxf(fmaxf(fs[0][0], fs[0][1]), fmaxf(fs[0][2], fs[0][3]));
    raw_m = fmaxf(fmaxf(fs[2][0], fs[2][1]), fmaxf(fs[2][2], fs[2][3]));
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < SC_NPT; ++i) {
        if (mx > WLX_NEG_INF) sm += __expf(w[i] - mx);
        if (nsp > 0 && raw_m > WLX_NEG_INF) raw_s += __expf(v[i] - raw_m);
    }
    sm = s3_wave_sum(sm);
    if (nsp > 0) raw_s = s3_wave_sum(raw_s);
    if (lane == 0) { fs[3][wave] = sm; fs[5][wave] = raw_s; }
    WLX_TR_MARK(2);

    // ---- candidate lists: each wave's top-ncand (no barrier), then wave 0 (and wave 1 for the timestamp-only list of
    // the chunk that straddles timestamp_begin) merges the four wave lists
    float* cv = st.scan_cv + ((long)r * (SC_MAXCH + 1) + chunk) * WLX_MAX_CAND;
    int* ci = st.scan_ci + ((long)r * (SC_MAXCH + 1) + chunk) * WLX_MAX_CAND;
    {
        // every lane's best AND second best of its 8 ids (ids ascend with i: the first maximum is the smallest id). A lane that wins a round
        // moves its second best up; only a lane that wins AGAIN rescans its ids (round 6: the rescan — ~60 instructions under one lane's exec
        // mask — ran after every round; 10 winners over 64 lanes mostly come from different lanes)
        float bv = WLX_NEG_INF, cv2 = WLX_NEG_INF; int bi = 0x7fffffff, ci2 = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < SC_NPT; ++i) {
            const int id = id0 + i * SC_THREADS + tid;
            if (w[i] > bv) { cv2 = bv; ci2 = bi; bv = w[i]; bi = id; }
            else if (w[i] > cv2) { cv2 = w[i]; ci2 = id; }
        }
        int won = 0;
#pragma unroll 1
        for (int k = 0; k < sp.ncand; ++k) {
            float gv = bv; int gi = bi;
            s3_wave_argmax(gv, gi);
            if (lane == 0) { wv_s[wave][k] = gv; wi_s[wave][k] = gi; }
            if (gi != 0x7fffffff && ((gi - id0) & (SC_THREADS - 1)) == tid) {
#pragma unroll
                for (int i = 0; i < SC_NPT; ++i)
                    if (id0 + i * SC_THREADS + tid == gi) w[i] = WLX_NEG_INF;        // the owner retires the winner
                if (won == 0) { bv = cv2; bi = ci2; }
                else {
                    // its next best: pairwise tree (depth 3) instead of a chain of 8; a tie keeps the LEFT operand = the smaller id
                    float tv[SC_NPT]; int ti_[SC_NPT];
#pragma unroll
                    for (int i = 0; i < SC_NPT; ++i) {
                        const int id = id0 + i * SC_THREADS + tid;
                        tv[i] = w[i]; ti_[i] = (w[i] > WLX_NEG_INF) ? id : 0x7fffffff;
                    }
#pragma unroll
                    for (int stp = 1; stp < SC_NPT; stp <<= 1)
#pragma unroll
                        for (int i = 0; i + stp < SC_NPT; i += 2 * stp)
                            if (tv[i + stp] > tv[i]) { tv[i] = tv[i + stp]; ti_[i] = ti_[i + stp]; }
                    bv = tv[0]; bi = ti_[0];
                }
                ++won;
            }
        }
        WLX_TR_MARK(3);
        __syncthreads();
        WLX_TR_MARK(4);
        if (wave == 0) {
            if (lane == 0) {
                float* so = st.scan_stats + ((long)r * SC_MAXCH + chunk) * SC_NSTAT;
                const float ssum = fs[3][0] + fs[3][1] + fs[3][2] + fs[3][3];
                so[0] = ts_chunk ? WLX_NEG_INF : mx; so[1] = ts_chunk ? 0.f : ssum;
                so[2] = ts_chunk ? mx : WLX_NEG_INF; so[3] = ts_chunk ? ssum : 0.f;
                so[4] = raw_m;   so[5] = fs[5][0] + fs[5][1] + fs[5][2] + fs[5][3];
            }
            // merge the four wave lists: <= 4 * ncand <= 128 candidates, two per lane, knock-out rounds (no rescans, no LDS)
            const int nc = sp.ncand;
            const int e0 = lane, e1 = lane + 64;
            float x0 = WLX_NEG_INF, x1 = WLX_NEG_INF; int i0 = 0x7fffffff, i1 = 0x7fffffff;
            if (e0 < 4 * nc) { const int q = e0 / nc, k = e0 - q * nc; x0 = wv_s[q][k]; i0 = wi_s[q][k]; }
            if (e1 < 4 * nc) { const int q = e1 / nc, k = e1 - q * nc; x1 = wv_s[q][k]; i1 = wi_s[q][k]; }
#pragma unroll 1
            for (int k = 0; k < nc; ++k) {
                const bool f1 = (x1 > x0) || (x1 == x0 && i1 < i0);
                float gv = f1 ? x1 : x0; int gi = f1 ? i1 : i0;
                s3_wave_argmax(gv, gi);
                if (lane == 0) { cv[k] = gv; ci[k] = gi; }
                if (gi != 0x7fffffff) {
                    if (i0 == gi) { x0 = WLX_NEG_INF; i0 = 0x7fffffff; }
                    if (i1 == gi) { x1 = WLX_NEG_INF; i1 = 0x7fffffff; }
                }
            }
        }
    }
    WLX_TR_MARK(5);
    WLX_TR_END(trc);
}

void launch_search_scan3(const float* logits, long ldl, int V, const SearchParams* sp_dev, int rows, const SearchState& st,
                         hipStream_t s) {
    hipLaunchKernelGGL(search_scan3_kernel, dim3(SC_MAXCH, rows), dim3(SC_THREADS), 0, s, logits, ldl, V, sp_dev, st WLX_TR_ARG("search_scan3"));
}

#define MU3_THREADS 384
// Round 6: the kernel was 11 us of ONE workgroup per step — five dependent round trips in front of its first phase (done flag -> parameters ->
// item state -> lists -> the row's cumulative score), and its bookkeeping as a serial loop of one thread over LDS (profiles/r6i_decode_step_trace.txt:
// 5.9 us to the end of phase 1, 5.1 us more for the beam merge and the loop). Now R (rows per item: part of the step graph's key) is a launch
// argument, so every address of the first phase is known at entry: the flag, the parameters, the item state AND the first row's chunk lists,
// statistics, cumulative score are requested together — one trip; the bookkeeping runs one candidate per lane (the serial loop's counters are
// prefix counts: ballots), the finished hypotheses' length penalties in parallel.
__global__ __launch_bounds__(MU3_THREADS) void search_merge_update3_kernel(const float* __restrict__ logits, long ldl, int V,
                                                                           const SearchParams* __restrict__ spp, SearchState st, int R WLX_TR_PARAM) {
    const int item = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = MU3_THREADS / 64;
    constexpr int PQ = 3;                        // list quads requested at entry: 12 candidates (beam <= 6); wider beams fetch the rest below
    const int r0 = item * R;
    // ---------------- everything phase 1 needs, requested before anything is used
    const int done_v = *st.done;
    const int idone = st.item_done[item];
    const int plen = st.plen[item];
    const int p = st.pos[r0];
    const int n_hyp0 = st.n_hyp[item];
    const int rp = r0 + (wave < R ? wave : R - 1);          // this wave's first row (the loop below runs it only if wave < beam <= R)
    const int lc = lane < SC_MAXCH ? lane : SC_MAXCH - 1;   // (lanes past the chunk count: any valid list — masked below)
    const float4* pso = reinterpret_cast<const float4*>(st.scan_stats + ((long)rp * SC_MAXCH + lc) * SC_NSTAT);
    const float4 ps0 = pso[0], ps1 = pso[1];
    const float4* pcv = reinterpret_cast<const float4*>(st.scan_cv + ((long)rp * (SC_MAXCH + 1) + lc) * WLX_MAX_CAND);
    const int4* pci = reinterpret_cast<const int4*>(st.scan_ci + ((long)rp * (SC_MAXCH + 1) + lc) * WLX_MAX_CAND);
    float4 pa[PQ]; int4 pb[PQ];
#pragma unroll
    for (int k = 0; k < PQ; ++k) { pa[k] = pcv[k]; pb[k] = pci[k]; }
    const float pcum = st.cum[rp];
    const int pnsp = st.nsp_row[rp];
    const SearchParams sp = *spp;
    if (done_v) return;
    WLX_TR_BEGIN();
    int step_no = 0;                             // (item 0, thread 0) this launch's number, 1-based (step_mirror)
    if (item == 0 && tid == 0) step_no = atomicAdd(st.step, 1) + 1;
    if (idone) { if (step_no) step_mirror(st, step_no); return; }
    int id0_, lim_, nct, nch;
    s3_chunk(0, sp.ts_begin, V, id0_, lim_, nct, nch);

    __shared__ float cand_s[16][WLX_MAX_CAND];
    __shared__ int cand_t[16][WLX_MAX_CAND];
    __shared__ float lv[NW][SC_MAXCH + 1][WLX_MAX_CAND];
    __shared__ int li[NW][SC_MAXCH + 1][WLX_MAX_CAND];
    __shared__ short anc_s[16 * WLX_T_TEXT];     // staged ancestry rows of this item (R <= 16)
    __shared__ int parent[16], newtok[16];
    __shared__ float newcum[16];
    __shared__ int hyp_src[16], hyp_extra[16], hyp_slot[16], hyp_n;
    __shared__ int n_active_s, finished_s;
    __shared__ int ord_b[WLX_MAX_CAND], ord_t[WLX_MAX_CAND];
    __shared__ float ord_s[WLX_MAX_CAND];
    __shared__ int4 rule_old[16];

    // the item's ancestry rows (old state) and rule state: requested now, used last
    for (int i = tid; i < sp.beam * (p + 1); i += MU3_THREADS) {
        const int b = i / (p + 1), q = i - b * (p + 1);
        anc_s[b * WLX_T_TEXT + q] = st.anc[(long)(r0 + b) * WLX_T_TEXT + q];
    }
    if (tid < sp.beam) rule_old[tid] = *reinterpret_cast<const int4*>(st.rule + 4 * (r0 + tid));

    // ---------------- phase 1: per beam row (one wave each), merge the chunk results into the row's ncand best
    const bool ts = sp.apply_ts_rules != 0;
    const int nc4 = (sp.ncand + 3) >> 2;
#pragma unroll 1
    for (int rb = wave; rb < sp.beam; rb += NW) {
        const int r = r0 + rb;
        const bool first = rb == wave;           // (wave-uniform) the row whose lists were requested at entry
        const bool have = lane < nch;
        float4 s0 = make_float4(WLX_NEG_INF, 0.f, WLX_NEG_INF, 0.f), s1 = make_float4(WLX_NEG_INF, 0.f, 0.f, 0.f);
        float cum_r; int nsp_r;
        {
            const float4* cv = reinterpret_cast<const float4*>(st.scan_cv + ((long)r * (SC_MAXCH + 1) + lc) * WLX_MAX_CAND);
            const int4* ci = reinterpret_cast<const int4*>(st.scan_ci + ((long)r * (SC_MAXCH + 1) + lc) * WLX_MAX_CAND);
            float4 a[WLX_MAX_CAND / 4]; int4 b[WLX_MAX_CAND / 4];
            if (first) {
#pragma unroll
                for (int k = 0; k < PQ; ++k) { a[k] = pa[k]; b[k] = pb[k]; }
#pragma unroll
                for (int k = PQ; k < WLX_MAX_CAND / 4; ++k)
                    if (k < nc4) { a[k] = cv[k]; b[k] = ci[k]; }
                if (have) { s0 = ps0; s1 = ps1; }
                cum_r = pcum; nsp_r = pnsp;
            } else {
#pragma unroll
                for (int k = 0; k < WLX_MAX_CAND / 4; ++k)
                    if (k < nc4) { a[k] = cv[k]; b[k] = ci[k]; }
                const float4* so = reinterpret_cast<const float4*>(st.scan_stats + ((long)r * SC_MAXCH + lc) * SC_NSTAT);
                const float4 t0 = so[0], t1 = so[1];
                if (have) { s0 = t0; s1 = t1; }
                cum_r = st.cum[r]; nsp_r = st.nsp_row[r];
            }
#pragma unroll
            for (int k = 0; k < WLX_MAX_CAND / 4; ++k)
                if (k < nc4 && have) {
                    *reinterpret_cast<float4*>(&lv[wave][lane][4 * k]) = have ? a[k] : make_float4(WLX_NEG_INF, WLX_NEG_INF, WLX_NEG_INF, WLX_NEG_INF);
                    *reinterpret_cast<int4*>(&li[wave][lane][4 * k]) = have ? b[k] : make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);
                }
        }
        const float mt = s0.x, st_ = s0.y, mts = s0.z, sts = s0.w, rm = s1.x, rs = s1.y;
        const float Mtext = s3_wave_max(mt), Mts = s3_wave_max(mts);
        if (nsp_r > 0) {
            const float RM = s3_wave_max(rm);
            const float RS = s3_wave_sum((rm > WLX_NEG_INF) ? rs * __expf(rm - RM) : 0.f);
            if (lane == 0) st.no_speech[item] = __expf(logits[(long)r * ldl + sp.no_speech] - RM) / RS;
        }
        const float S_ts = s3_wave_sum((mts > WLX_NEG_INF) ? sts * __expf(mts - Mts) : 0.f);
        const float lse_ts = (Mts > WLX_NEG_INF) ? Mts + __logf(S_ts) : WLX_NEG_INF;
        const bool text_masked = ts && (lse_ts > Mtext);
        float lse_sel;
        if (text_masked) {
            lse_sel = lse_ts;
        } else {
            const float mx_all = fmaxf(Mtext, Mts);
            const float a = ((mt > WLX_NEG_INF) ? st_ * __expf(mt - mx_all) : 0.f) +
                            ((mts > WLX_NEG_INF) ? sts * __expf(mts - mx_all) : 0.f);
            lse_sel = mx_all + __logf(s3_wave_sum(a));
        }
        const bool usable = (lane < nch) && (!text_masked || lane >= nct);   // timestamps only: the timestamp chunks
        const float base = cum_r - lse_sel;
        int hp = 0;
        float hv = usable ? lv[wave][lane][0] : WLX_NEG_INF;
        int hi = usable ? li[wave][lane][0] : 0x7fffffff;
#pragma unroll 1
        for (int k = 0; k < sp.ncand; ++k) {
            float gv = hv; int gi = hi;
            s3_wave_argmax(gv, gi);
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

    // ---------------- phase 2: merge the beams' sorted lists (score desc, row asc, list position asc) into ord_*
    const int ngen = p + 1 - plen;
    const int max_new = sp.max_length - plen;
    const bool is_last = (ngen + 1 >= max_new);
    if (wave == 0) {
        // lane b < beam (<= 16: one DPP row) owns row b's sorted list; round winner = (score desc, b asc), which is the
        // serial scan's strict '>' over ascending b. A lane with nothing left ranks below every real entry, including
        // a real -inf score: it ties on value and loses on b.
        const int b = lane;
        const bool own = b < sp.beam;
        int hp = 0;
#pragma unroll 1
        for (int k = 0; k < sp.ncand; ++k) {
            const bool alive = own && hp < sp.ncand;
            float kv = alive ? cand_s[b][hp] : WLX_NEG_INF;
            int kb = alive ? b : 0x7fffffff;
            s3_row_argmax(kv, kb);
            if (lane == 0) ord_b[k] = (kb == 0x7fffffff) ? -1 : kb;
            if (alive && b == kb) { ord_s[k] = cand_s[b][hp]; ord_t[k] = cand_t[b][hp]; ++hp; }
        }
        // ---------------- bookkeeping, one merged candidate per lane (k < ncand <= 32). The serial statement of it (search_update_kernel): walk
        // k upwards, stop at the first empty entry; a finished candidate (EOT, or the last step) with k < beam becomes hypothesis number
        // n_hyp + (finished ones accepted so far) while slots last; any other candidate becomes active row number (others so far) while
        // fewer than beam are active. Both counters only grow with k, so "so far" is a prefix count over the lanes below.
        const int k = lane;
        const bool in_list = k < sp.ncand;
        const int bb = in_list ? ord_b[k] : -1;                       // (LDS written by this wave's lanes above: same-wave program order)
        const float bs = in_list ? ord_s[k] : 0.f;
        const int tok = in_list ? ord_t[k] : 0;
        const unsigned long long stop = __ballot(!in_list || bb < 0);         // never empty: lane ncand <= 32 is set
        const bool live = k < (int)__builtin_ctzll(stop);
        const bool fin_k = live && (tok == sp.eot || is_last);
        const bool act_k = live && !fin_k;
        const unsigned long long below = (1ull << k) - 1ull;
        const unsigned long long finm = __ballot(fin_k && k < sp.beam);
        const unsigned long long actm = __ballot(act_k);
        int cap = WLX_MAX_HYP - n_hyp0; cap = cap < 16 ? cap : 16; cap = cap > 0 ? cap : 0;
        const int hidx = __builtin_popcountll(finm & below), aidx = __builtin_popcountll(actm & below);
        if (fin_k && k < sp.beam && hidx < cap) {
            const int len = ngen + ((tok == sp.eot) ? 0 : 1);
            const int slot = n_hyp0 + hidx;
            hyp_src[hidx] = bb;
            hyp_extra[hidx] = (tok == sp.eot) ? -1 : tok;
            hyp_slot[hidx] = slot;
            st.hyp_len[item * WLX_MAX_HYP + slot] = len;
            const float flen = (float)(len > 0 ? len : 1);
            const float denom = (sp.length_penalty == 1.0f) ? flen : powf(flen, sp.length_penalty);    // (pow(x, 1) = x exactly: the default penalty skips ~200 instructions)
            st.hyp_score[item * WLX_MAX_HYP + slot] = bs / denom;
        }
        if (act_k && aidx < sp.beam) { parent[aidx] = bb; newtok[aidx] = tok; newcum[aidx] = bs; }
        if (lane == 0) {
            int nh_new = __builtin_popcountll(finm); nh_new = nh_new < cap ? nh_new : cap;
            int n_active = __builtin_popcountll(actm); n_active = n_active < sp.beam ? n_active : sp.beam;
            const bool top_beam_finished = fin_k;                     // (lane 0 = candidate 0)
            const int n_hyp = n_hyp0 + nh_new;
            st.n_hyp[item] = n_hyp; st.n_hyp_host[item] = n_hyp;
            hyp_n = nh_new;
            n_active_s = n_active;
            bool fin = is_last || n_active == 0;
            if (sp.allow_early_exit) fin = fin || (top_beam_finished && n_hyp >= sp.num_hyp);
            else fin = fin || (n_hyp >= sp.max_cand_hyp);
            finished_s = fin ? 1 : 0;
        }
    }
    __syncthreads();
    WLX_TR_MARK(2);
    for (int hh = 0; hh < hyp_n; ++hh) {
        const int b = hyp_src[hh];
        int* dst = st.hyp_tokens + ((long)item * WLX_MAX_HYP + hyp_slot[hh]) * WLX_T_TEXT;
        for (int j = tid; j < ngen; j += MU3_THREADS)
            dst[j] = st.intok[(long)anc_s[b * WLX_T_TEXT + plen + j] * WLX_T_TEXT + plen + j];
        if (tid == 0 && hyp_extra[hh] >= 0) dst[ngen] = hyp_extra[hh];
    }
    if (finished_s) {
        __syncthreads();                                 // every thread's hypothesis tokens are stored
        if (tid == 0) {
            if (step_no) step_mirror(st, step_no);       // (before the done word: the host may start its next call the moment it sees that one)
            finish_item(st, item, sp.items);
        }
        return;
    }
    const int na = n_active_s;
    for (int i = tid; i < na * (p + 1); i += MU3_THREADS) {
        const int j = i / (p + 1), q = i - j * (p + 1);
        st.anc[(long)(r0 + j) * WLX_T_TEXT + q] = anc_s[parent[j] * WLX_T_TEXT + q];
    }
    if (tid < sp.beam) {
        const int j = tid;
        st.anc[(long)(r0 + j) * WLX_T_TEXT + p + 1] = (short)(r0 + j);
        int4 ru = make_int4(0, 1, -1, 0);
        if (j < na) {
            const int tok = newtok[j];
            const int4 po = rule_old[parent[j]];
            const int is_ts = tok >= sp.ts_begin ? 1 : 0;
            // after this token: last = it; one-before-last = the parent's last (none yet when this is the first token)
            ru.x = is_ts;
            ru.y = (ngen + 1 < 2) ? 1 : po.x;
            ru.z = is_ts ? tok : po.z;
            st.token[r0 + j] = tok; st.cum[r0 + j] = newcum[j];
        } else { st.token[r0 + j] = sp.eot; st.cum[r0 + j] = WLX_NEG_INF; }
        *reinterpret_cast<int4*>(st.rule + 4 * (r0 + j)) = ru;
        st.pos[r0 + j] = p + 1;
        st.nsp_row[r0 + j] = 0;
    }
    if (step_no) step_mirror(st, step_no);
    WLX_TR_END(trc);
}

void launch_search_merge_update3(const float* logits, long ldl, int V, const SearchParams* sp_dev, int items, int R,
                                 const SearchState& st, hipStream_t s) {
    hipLaunchKernelGGL(search_merge_update3_kernel, dim3(items), dim3(MU3_THREADS), 0, s, logits, ldl, V, sp_dev, st, R WLX_TR_ARG("search_merge_update3"));
}

// ------------------------------------------------------------------ per-call reset of the search state (one launch
// instead of eight hipMemsetAsync calls in front of every wlx_generate: ~5 us of host time each)
__global__ __launch_bounds__(256) void search_reset_kernel(SearchState st, int items, int rows) {
    const int t = threadIdx.x;
    if (t == 0) { *st.step = 0; *st.done = 0; *st.n_finished = 0; step_mirror(st, 0); }   // (the mirror word too: stream-ordered behind every kernel of the previous call)
    for (int i = t; i < items; i += 256) { st.item_done[i] = 0; st.n_hyp[i] = 0; st.n_hyp_host[i] = 0; st.no_speech[i] = 0.f; }
    for (int i = t; i < rows; i += 256) st.row_done[i] = 0;
    for (int i = t; i < items * WLX_MAX_HYP; i += 256) st.hyp_len[i] = 0;
}
void launch_search_reset(const SearchState& st, int items, int rows, hipStream_t s) {
    hipLaunchKernelGGL(search_reset_kernel, dim3(1), dim3(256), 0, s, st, items, rows);
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
__global__ __launch_bounds__(SR_THREADS) void token_prob_rows_kernel(const float* __restrict__ logits, long ldl, int vlim,
                                                                     const int* __restrict__ toks, float* __restrict__ out) {
    __shared__ float fs[16];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* lrow = logits + (long)r * ldl;
    float mx = WLX_NEG_INF;
    for (int i = tid; i < vlim; i += SR_THREADS) mx = fmaxf(mx, lrow[i]);
    mx = block_max(mx, fs);
    float sm = 0.f;
    for (int i = tid; i < vlim; i += SR_THREADS) sm += __expf(lrow[i] - mx);
    sm = block_sum(sm, fs);
    if (tid == 0) { const int t = toks[r]; out[r] = (t >= 0 && t < vlim) ? __expf(lrow[t] - mx) / sm : 0.f; }
}
void launch_token_prob_rows(const float* logits, long ldl, int vlim, int rows, const int* toks, float* out, hipStream_t s) {
    hipLaunchKernelGGL(token_prob_rows_kernel, dim3(rows), dim3(SR_THREADS), 0, s, logits, ldl, vlim, toks, out);
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
