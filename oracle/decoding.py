"""CPU oracle for the token search inside ctranslate2.models.Whisper.generate (numpy).
Test infrastructure only — see oracle/__init__.py.

The reference drives it at whisper_live/transcriber/transcriber_faster_whisper.py:1380-1407 (T=0: beam_size=5,
patience=1; T>0: beam_size=1, num_hypotheses=best_of, sampling_topk=0, sampling_temperature=T) and consumes
``sequences_ids[0]`` (prompt and EOT excluded), ``scores[0]`` (= sum of log-probs incl. EOT / len^length_penalty,
recovered at :1412-1414) and ``no_speech_prob``. CTranslate2's source is not in the reference tree, so:

* logits processors follow the published OpenAI definition (whisper/decoding.py SuppressBlank, SuppressTokens,
  ApplyTimestampRules) == transformers generation/logits_process.py:1909-2047 (WhisperTimeStampLogitsProcessor),
  which tests/ cross-check against;
* beam search follows the CT2 contract of SURVEY.md Appendix A.5: 2*beam candidates per step ranked by cumulative
  log-prob; a candidate ending in EOT inside the top `beam` becomes a finished hypothesis; the first `beam`
  non-EOT candidates continue; the search stops once round(beam*patience) hypotheses are finished (or, with
  length_penalty == 0, as soon as the best candidate is finished), or at max_length; hypotheses are ranked by
  sum_logp / len^length_penalty. Ties: higher score first, then lower token id / lower beam index.
  CT2's own tie-breaking and RNG stream are parity-unpinned.

This file and whisperlive_amd/csrc/search.hip implement the SAME decision procedure.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

NEG_INF = np.float32(-np.inf)


@dataclass
class TokenIds:
    sot: int
    eot: int
    no_timestamps: int
    timestamp_begin: int
    no_speech: int
    blank: int = -1


@dataclass
class GenOptions:
    ids: TokenIds
    beam_size: int = 5
    patience: float = 1.0
    num_hypotheses: int = 1
    length_penalty: float = 1.0
    repetition_penalty: float = 1.0
    no_repeat_ngram_size: int = 0
    max_length: int = 448
    suppress_blank: bool = True
    suppress_tokens: Sequence[int] = ()
    max_initial_timestamp_index: int = 50
    sampling_topk: int = 0
    sampling_temperature: float = 0.0
    seed: int = 0


@dataclass
class GenResult:
    sequences_ids: List[List[int]]
    scores: List[float]
    no_speech_prob: float
    steps: int = 0


def _lse(v: np.ndarray) -> np.float32:
    m = v.max()
    if not np.isfinite(m):
        return NEG_INF
    return np.float32(m + np.log(np.exp((v - m).astype(np.float32)).sum(dtype=np.float32)))


def process_logits(logits: np.ndarray, history: Sequence[int], o: GenOptions, apply_ts: bool
                   ) -> Tuple[np.ndarray, np.float32, np.float32]:
    """Apply the logits processors to ONE row. Returns (masked logits, lse over the surviving set, its max)."""
    ids = o.ids
    v = np.array(logits, dtype=np.float32, copy=True)
    V = v.shape[0]
    ngen = len(history)
    if o.repetition_penalty != 1.0 and ngen:
        seen = np.unique(np.asarray(history))
        sel = v[seen]
        v[seen] = np.where(sel < 0, sel * np.float32(o.repetition_penalty), sel / np.float32(o.repetition_penalty))
    n = o.no_repeat_ngram_size
    if n > 0 and ngen >= n - 1:
        tail = list(history[ngen - (n - 1):]) if n > 1 else []
        for j in range(0, ngen - n + 1):
            if list(history[j:j + n - 1]) == tail:
                v[history[j + n - 1]] = NEG_INF
    if len(o.suppress_tokens):
        st = np.asarray([t for t in o.suppress_tokens if 0 <= t < V], dtype=np.int64)
        v[st] = NEG_INF
    first = ngen == 0
    if first and o.suppress_blank:
        if ids.blank >= 0:
            v[ids.blank] = NEG_INF
        v[ids.eot] = NEG_INF
    text_masked = False
    if apply_ts:
        tb = ids.timestamp_begin
        v[ids.no_timestamps] = NEG_INF
        last_was_ts = ngen >= 1 and history[-1] >= tb
        penult_was_ts = ngen < 2 or history[-2] >= tb
        if last_was_ts:
            if penult_was_ts:
                v[tb:] = NEG_INF
            else:
                v[: ids.eot] = NEG_INF
        ts_hist = [t for t in history if t >= tb]
        if ts_hist:
            ts_last = ts_hist[-1] if (last_was_ts and not penult_was_ts) else ts_hist[-1] + 1
            v[tb:ts_last] = NEG_INF
        if first:
            v[:tb] = NEG_INF
            if o.max_initial_timestamp_index >= 0:
                v[tb + o.max_initial_timestamp_index + 1:] = NEG_INF
        lse_ts = _lse(v[tb:])
        mx_text = v[:tb].max() if tb > 0 else NEG_INF
        if lse_ts > mx_text:          # same comparison in logit space as in log-prob space
            v[:tb] = NEG_INF
            text_masked = True
    lse = _lse(v)
    return v, lse, np.float32(v.max())


class LogitsProvider:
    """What the search needs from the network: next-token logits for the live rows."""

    def prefill(self, tokens: Sequence[int]) -> Optional[np.ndarray]:
        """Feed prompt[:-1]; returns logits [len, V] of those positions (or None if unavailable)."""
        raise NotImplementedError

    def step(self, tokens: Sequence[int], parents: Sequence[int]) -> np.ndarray:
        """Reorder the rows by `parents`, feed one token per row, return logits [rows, V]."""
        raise NotImplementedError


class InjectedLogits(LogitsProvider):
    def __init__(self, logits: np.ndarray):
        self.logits = logits  # [steps, rows, V]
        self.i = 0

    def prefill(self, tokens):
        return None

    def step(self, tokens, parents):
        out = self.logits[self.i][: len(tokens)]
        self.i += 1
        return out


_M64 = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def uniform01(seed: int, row: int, pos: int) -> np.float32:
    h = _splitmix64((seed & _M64) ^ _splitmix64(((row << 32) | (pos & 0xFFFFFFFF)) & _M64))
    return np.float32(h >> 40) * np.float32(1.0 / 16777216.0)


def sample_token(v: np.ndarray, mx: np.float32, temperature: float, u: np.float32, eot: int, chunk: int = 52) -> int:
    """Inverse-CDF draw over softmax((v - mx)/T) in id order, float32, chunked exactly like search.hip."""
    V = v.shape[0]
    with np.errstate(invalid="ignore"):
        e = np.exp(((v - mx) * np.float32(1.0 / temperature)).astype(np.float32)).astype(np.float32)
    e[~np.isfinite(v)] = 0.0
    nchunks = 1024
    sums = np.zeros(nchunks, dtype=np.float32)
    for c in range(nchunks):
        seg = e[c * chunk:(c + 1) * chunk]
        acc = np.float32(0.0)
        for x in seg:
            acc = np.float32(acc + x)
        sums[c] = acc
    total = np.float32(0.0)
    for c in range(nchunks):
        total = np.float32(total + sums[c])
    target = np.float32(u * total)
    run = np.float32(0.0)
    ch = 0
    while ch < nchunks - 1:
        if np.float32(run + sums[ch]) > target:
            break
        run = np.float32(run + sums[ch])
        ch += 1
    pick, lastvalid = -1, -1
    for q in range(chunk):
        i = ch * chunk + q
        if i >= V:
            break
        if e[i] > 0:
            lastvalid = i
        run = np.float32(run + e[i])
        if run > target and e[i] > 0:
            pick = i
            break
    if pick < 0:
        pick = lastvalid if lastvalid >= 0 else eot
    return int(pick)


def generate(provider: LogitsProvider, prompt: Sequence[int], o: GenOptions, row_base: int = 0) -> GenResult:
    """One audio item. Mirrors wlx_generate / search.hip."""
    ids = o.ids
    prompt = list(prompt)
    plen = len(prompt)
    assert 1 <= plen < o.max_length
    apply_ts = ids.no_timestamps not in prompt
    sampling = o.sampling_temperature > 0 or o.beam_size <= 1
    max_new = o.max_length - plen
    no_speech = 0.0
    sot_idx = max((i for i, t in enumerate(prompt) if t == ids.sot), default=-1)
    pre = provider.prefill(prompt[:-1]) if plen > 1 else None
    if pre is not None and 0 <= sot_idx < plen - 1:
        row = pre[sot_idx].astype(np.float32)
        no_speech = float(np.exp(row[ids.no_speech] - _lse(row)))

    if not sampling:
        B = o.beam_size
        ncand = 2 * B
        max_hyp = max(1, int(round(B * o.patience)))
        allow_early_exit = o.length_penalty == 0
        cum = np.full(B, NEG_INF, dtype=np.float32)
        cum[0] = 0.0
        hist: List[List[int]] = [[] for _ in range(B)]
        feed = [prompt[-1]] * B
        parents = [0] * B
        hyps: List[Tuple[float, List[int]]] = []
        steps = 0
        for step in range(max_new):
            logits = provider.step(feed, parents)
            steps += 1
            if step == 0 and sot_idx == plen - 1:
                row = np.asarray(logits[0], dtype=np.float32)
                no_speech = float(np.exp(row[ids.no_speech] - _lse(row)))
            cands = []  # (score, row, rank, token)
            for b in range(B):
                v, lse, _ = process_logits(logits[b], hist[b], o, apply_ts)
                base = np.float32(cum[b] - lse)
                order = np.lexsort((np.arange(v.shape[0]), -v))[:ncand]   # value desc, id asc
                for rank, tok in enumerate(order):
                    sc = np.float32(v[tok] + base) if np.isfinite(v[tok]) else NEG_INF
                    cands.append((sc, b, rank, int(tok)))
            with np.errstate(invalid="ignore"):
                cands.sort(key=lambda c: (-c[0] if not np.isnan(c[0]) else np.inf, c[1], c[2]))
            cands = cands[:ncand]
            is_last = step + 1 >= max_new
            new_parent, new_tok, new_cum = [], [], []
            top_finished = False
            for k, (sc, b, rank, tok) in enumerate(cands):
                if tok == ids.eot or is_last:
                    if k >= B:
                        continue
                    toks = list(hist[b]) + ([] if tok == ids.eot else [tok])
                    denom = np.float32(max(len(toks), 1)) ** np.float32(o.length_penalty)
                    hyps.append((float(np.float32(sc) / denom), toks))
                    if k == 0:
                        top_finished = True
                elif len(new_parent) < B:
                    new_parent.append(b); new_tok.append(tok); new_cum.append(sc)
            fin = is_last or not new_parent
            if allow_early_exit:
                fin = fin or (top_finished and len(hyps) >= o.num_hypotheses)
            else:
                fin = fin or len(hyps) >= max_hyp
            if fin:
                break
            hist = [hist[p] + [t] for p, t in zip(new_parent, new_tok)]
            while len(hist) < B:   # (cannot happen with >= beam non-EOT candidates; kept for symmetry)
                hist.append(list(hist[-1])); new_parent.append(new_parent[-1]); new_tok.append(ids.eot); new_cum.append(NEG_INF)
            cum = np.asarray(new_cum, dtype=np.float32)
            feed, parents = new_tok, new_parent
        order = sorted(range(len(hyps)), key=lambda i: -hyps[i][0])   # stable: insertion order on ties
        order = order[: o.num_hypotheses]
        return GenResult([hyps[i][1] for i in order], [hyps[i][0] for i in order], no_speech, steps)

    # ---- sampling / greedy: num_hypotheses independent rows
    R = max(1, o.num_hypotheses)
    cum = np.zeros(R, dtype=np.float32)
    hist = [[] for _ in range(R)]
    feed = [prompt[-1]] * R
    done = [False] * R
    out: List[Optional[Tuple[float, List[int]]]] = [None] * R
    steps = 0
    for step in range(max_new):
        logits = provider.step(feed, list(range(R)))
        steps += 1
        if step == 0 and sot_idx == plen - 1:
            row = np.asarray(logits[0], dtype=np.float32)
            no_speech = float(np.exp(row[ids.no_speech] - _lse(row)))
        is_last = step + 1 >= max_new
        for r in range(R):
            if done[r]:
                continue
            v, lse, mx = process_logits(logits[r], hist[r], o, apply_ts)
            if o.sampling_temperature <= 0 or o.sampling_topk == 1:
                tok = int(np.lexsort((np.arange(v.shape[0]), -v))[0])
            else:
                u = uniform01(o.seed, row_base + r, plen - 1 + step)
                tok = sample_token(v, mx, o.sampling_temperature, u, ids.eot)
            cum[r] = np.float32(cum[r] + np.float32(v[tok] - lse))
            if tok == ids.eot or is_last:
                toks = hist[r] + ([] if tok == ids.eot else [tok])
                denom = np.float32(max(len(toks), 1)) ** np.float32(o.length_penalty)
                out[r] = (float(cum[r] / denom), toks)
                done[r] = True
            else:
                hist[r] = hist[r] + [tok]
                feed[r] = tok
        if all(done):
            break
    res = [x for x in out if x is not None]
    order = sorted(range(len(res)), key=lambda i: -res[i][0])
    return GenResult([res[i][1] for i in order], [res[i][0] for i in order], no_speech, steps)
