"""Word-level timestamps on top of the engine's alignment entry point (wlx_align): the host half of
``WhisperModel.add_word_timestamps`` / ``find_alignment`` / ``merge_punctuations``
(whisper_live/transcriber/transcriber_faster_whisper.py:1515-1714, 1856-1887) and of the hallucination-silence rules
that only exist with word timings (:1186-1291). Pure bookkeeping over (text index, time index) paths — the attention
scores, token probabilities and the DTW come from libwlx."""
from __future__ import annotations


from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

SENTENCE_END = ".。!！?？"
# the reference's own list for the anomaly score (:1078; a SUBSTRING test against this string, not a set of characters)
ANOMALY_PUNCTUATION = "\"'“¿([{-\"'.。,，!！?？:：”)]}、"


def last_word_end(segments: List[dict]) -> Optional[float]:
    """faster_whisper.utils.get_end: end of the last word of the last segment that has words, else the last segment's end"""
    for seg in reversed(segments):
        for w in reversed(seg.get("words") or []):
            return w["end"]
    return segments[-1]["end"] if segments else None


def merge_punctuations(alignment: List[dict], prepended: str, appended: str) -> None:
    """glue opening punctuation to the word that follows and closing punctuation to the word before (:1856-1887);
    absorbed entries stay in the list with an empty word so token bookkeeping is unchanged"""
    nxt = len(alignment) - 1
    for i in range(len(alignment) - 2, -1, -1):
        cur, fol = alignment[i], alignment[nxt]
        if cur["word"].startswith(" ") and cur["word"].strip() in prepended:
            fol["word"] = cur["word"] + fol["word"]
            fol["tokens"] = cur["tokens"] + fol["tokens"]
            cur["word"], cur["tokens"] = "", []
        else:
            nxt = i
    prev = 0
    for j in range(1, len(alignment)):
        cur, fol = alignment[prev], alignment[j]
        if not cur["word"].endswith(" ") and fol["word"] in appended:
            cur["word"] = cur["word"] + fol["word"]
            cur["tokens"] = cur["tokens"] + fol["tokens"]
            fol["word"], fol["tokens"] = "", []
        else:
            prev = j


def words_from_path(tokenizer, text_tokens: Sequence[int], text_indices: np.ndarray, time_indices: np.ndarray,
                    text_token_probs: np.ndarray, tokens_per_second: int) -> List[dict]:
    """(:1665-1714) word strings / token groups from the tokenizer, word start and end from the first time index at
    which the DTW path enters the word's first token and the following word's first token"""
    words, groups = tokenizer.split_to_word_tokens(list(text_tokens) + [tokenizer.eot])
    if len(groups) <= 1:
        return []
    bounds = np.concatenate([[0], np.cumsum([len(g) for g in groups[:-1]])]).astype(np.int64)
    if len(bounds) <= 1:
        return []
    enters = np.concatenate([[True], np.diff(text_indices) != 0])
    enter_times = time_indices[enters] / tokens_per_second
    starts, ends = enter_times[bounds[:-1]], enter_times[bounds[1:]]
    probs = [float(np.mean(text_token_probs[a:b])) for a, b in zip(bounds[:-1], bounds[1:])]
    return [dict(word=w, tokens=g, start=float(s), end=float(e), probability=p)
            for w, g, s, e, p in zip(words, groups, starts, ends, probs)]


def add_word_timestamps(segments: List[List[dict]], tokenizer, align_fn: Callable, num_frames: int, tokens_per_second: int,
                        frames_per_second: int, prepend_punctuations: str, append_punctuations: str,
                        last_speech_timestamp: float) -> Optional[float]:
    """(:1515-1644) `segments` = one list of sub-segment dicts per encoded window; align_fn(text_tokens, num_frames, window)
    -> (text_indices, time_indices, text_token_probs). Fills sub-segment["words"], may move sub-segment start / end, and
    returns the updated last-speech timestamp."""
    if not segments:
        return None
    per_window_tokens = []
    alignments = []
    for wi, window in enumerate(segments):
        per_sub = [[t for t in sub["tokens"] if t < tokenizer.eot] for sub in window]
        per_window_tokens.append(per_sub)
        flat = [t for toks in per_sub for t in toks]
        if not flat:
            alignments.append([])
            continue
        ti, fi, probs = align_fn(flat, num_frames, wi)
        alignments.append(words_from_path(tokenizer, flat, ti, fi, probs, tokens_per_second))
    limits = []
    for al in alignments:
        durs = np.array([w["end"] - w["start"] for w in al])
        durs = durs[durs.nonzero()]
        med = min(0.7, float(np.median(durs))) if len(durs) else 0.0
        mx = med * 2
        if len(durs):
            # words at sentence boundaries must not be longer than twice the median word duration
            for i in range(1, len(al)):
                if al[i]["end"] - al[i]["start"] > mx:
                    if al[i]["word"] in SENTENCE_END:
                        al[i]["end"] = al[i]["start"] + mx
                    elif al[i - 1]["word"] in SENTENCE_END:
                        al[i]["start"] = al[i]["end"] - mx
        merge_punctuations(al, prepend_punctuations, append_punctuations)
        limits.append((med, mx))
    for wi, window in enumerate(segments):
        if not window:
            continue
        al = alignments[wi]
        med, mx = limits[wi]
        offset = window[0]["seek"] / frames_per_second
        wpos = 0
        for si, sub in enumerate(window):
            n_sub = len(per_window_tokens[wi][si])
            taken = 0
            words = []
            while wpos < len(al) and taken < n_sub:
                t = al[wpos]
                if t["word"]:
                    words.append(dict(word=t["word"], start=round(offset + t["start"], 2), end=round(offset + t["end"], 2),
                                      probability=t["probability"]))
                taken += len(t["tokens"])
                wpos += 1
            if words:
                # the first (and second) word after a pause must not be longer than twice the median duration
                if words[0]["end"] - last_speech_timestamp > med * 4 and (
                        words[0]["end"] - words[0]["start"] > mx
                        or (len(words) > 1 and words[1]["end"] - words[0]["start"] > mx * 2)):
                    if len(words) > 1 and words[1]["end"] - words[1]["start"] > mx:
                        cut = max(words[1]["end"] / 2, words[1]["end"] - mx)
                        words[0]["end"] = words[1]["start"] = cut
                    words[0]["start"] = max(0, words[0]["end"] - mx)
                # prefer the segment-level start if the first word is too long
                if sub["start"] < words[0]["end"] and sub["start"] - 0.5 > words[0]["start"]:
                    words[0]["start"] = max(0, min(words[0]["end"] - med, sub["start"]))
                else:
                    sub["start"] = words[0]["start"]
                # prefer the segment-level end if the last word is too long
                if sub["end"] > words[-1]["start"] and sub["end"] + 0.5 < words[-1]["end"]:
                    words[-1]["end"] = max(words[-1]["start"] + med, sub["end"])
                else:
                    sub["end"] = words[-1]["end"]
                last_speech_timestamp = sub["end"]
            sub["words"] = words
    return last_speech_timestamp


# ---- hallucination heuristics that need word timings (:1186-1210)
def word_anomaly_score(word: dict) -> float:
    """very improbable, very short or very long words"""
    dur = word["end"] - word["start"]
    score = 1.0 if word.get("probability", 0.0) < 0.15 else 0.0
    if dur < 0.133:
        score += (0.133 - dur) * 15
    if dur > 2.0:
        score += dur - 2.0
    return score


def is_segment_anomaly(segment: Optional[dict]) -> bool:
    if segment is None or not segment["words"]:
        return False
    words = [w for w in segment["words"] if w["word"] not in ANOMALY_PUNCTUATION][:8]
    score = sum(word_anomaly_score(w) for w in words)
    return score >= 3 or score + 0.01 >= len(words)


def next_words_segment(segments: List[dict]) -> Optional[dict]:
    return next((s for s in segments if s["words"]), None)


def default_alignment_heads(dec_layers: int, n_heads: int) -> List[Tuple[int, int]]:
    """without a model-specific list (generation_config.json "alignment_heads") the published default is every head
    of the upper half of the decoder"""
    return [(l, h) for l in range(dec_layers // 2, dec_layers) for h in range(n_heads)]
