"""CPU tests of the word-timestamp path: the oracle's median filter / DTW against the Hugging Face golden vectors
(tests/golden/align_golden.npz, made by tests/golden/make_align_golden.py), the tokenizer's word splitting, and the
host bookkeeping of whisperlive_amd/word_timing.py (reference: transcriber_faster_whisper.py:1515-1714, 1856-1887)."""
import os

import numpy as np

from oracle import alignment as oal
from whisperlive_amd import word_timing as wt

GOLD = os.path.join(os.path.dirname(__file__), "golden", "align_golden.npz")


def test_oracle_dtw_and_median_filter_match_hf_golden():
    g = np.load(GOLD)
    for i in range(4):
        ti, fi = oal.dtw(g[f"dtw_in_{i}"])
        assert np.array_equal(ti, g[f"dtw_ti_{i}"]) and np.array_equal(fi, g[f"dtw_fi_{i}"])
        # a DTW path is monotone, starts at (0, 0) and ends at (N-1, M-1)
        assert ti[0] == 0 and fi[0] == 0 and ti[-1] == g[f"dtw_in_{i}"].shape[0] - 1 and fi[-1] == g[f"dtw_in_{i}"].shape[1] - 1
        assert (np.diff(ti) >= 0).all() and (np.diff(fi) >= 0).all()
        y = oal.median_filter(g[f"med_in_{i}"], int(g[f"med_w_{i}"]))
        assert np.array_equal(y, g[f"med_out_{i}"])


def test_merge_punctuations():
    al = [dict(word=" (", tokens=[1]), dict(word="hello", tokens=[2, 3]), dict(word=",", tokens=[4]),
          dict(word=" world", tokens=[5]), dict(word="!", tokens=[6]), dict(word=")", tokens=[7])]
    wt.merge_punctuations(al, "\"'“¿([{-", "\"'.。,，!！?？:：”)]}、")
    assert [a["word"] for a in al] == ["", " (hello,", "", " world!)", "", ""]
    assert [a["tokens"] for a in al] == [[], [1, 2, 3, 4], [], [5, 6, 7], [], []]
    assert sum(len(a["tokens"]) for a in al) == 7            # token bookkeeping is preserved


class _Tok:
    """words = runs of ids; id >= 100 is 'special'; a word starts at every id divisible by 10"""
    eot = 100
    language_code = "en"

    def split_to_word_tokens(self, tokens):
        words, groups = [], []
        for t in tokens:
            if t >= self.eot or t % 10 == 0 or not groups:
                words.append(f" w{t}"); groups.append([t])
            else:
                words[-1] += f"+{t}"; groups[-1].append(t)
        return words, groups


def test_words_from_path_and_add_word_timestamps():
    tok = _Tok()
    text = [10, 11, 20, 30, 31, 32]                       # three words: [10,11] [20] [30,31,32]
    # path over rows (no_timestamps + 6 text tokens): row r is entered at frame 5 r (2 frames = one 20 ms step)
    ti = np.repeat(np.arange(7), 5)
    fi = np.arange(35)
    probs = np.array([0.9, 0.7, 0.5, 0.2, 0.4, 0.6], np.float32)
    words = wt.words_from_path(tok, text, ti, fi, probs, tokens_per_second=50)
    assert [w["tokens"] for w in words] == [[10, 11], [20], [30, 31, 32]]
    np.testing.assert_allclose([w["start"] for w in words], [0.0, 0.2, 0.3])
    np.testing.assert_allclose([w["end"] for w in words], [0.2, 0.3, 0.6])
    np.testing.assert_allclose([w["probability"] for w in words], [0.8, 0.5, 0.4], rtol=1e-6)

    subs = [dict(seek=300, start=3.0, end=3.25, tokens=[10, 11, 20, 150]), dict(seek=300, start=3.25, end=3.7, tokens=[151, 30, 31, 32, 152])]
    last = wt.add_word_timestamps([subs], tok, lambda tt, nf, w: (ti, fi, probs), 3000, 50, 100, "\"'“¿([{-", "\"'.。,，!！?？:：”)]}、", 0.0)
    assert [[w["word"] for w in s["words"]] for s in subs] == [[" w10+11", " w20"], [" w30+31+32"]]
    assert subs[0]["words"][0]["start"] == 3.0 and subs[0]["words"][1]["end"] == 3.3      # seek offset 300 frames = 3 s
    assert subs[0]["end"] == 3.3 and subs[1]["start"] == 3.3 and subs[1]["end"] == 3.6    # segment bounds follow the words
    assert last == 3.6 and wt.last_word_end(subs) == 3.6


def test_anomaly_rules():
    ok = dict(words=[dict(word=" a", start=0.0, end=0.3, probability=0.9)] * 4)
    bad = dict(words=[dict(word=" a", start=0.0, end=0.01, probability=0.05)] * 4)      # improbable and 10 ms long
    assert not wt.is_segment_anomaly(ok) and wt.is_segment_anomaly(bad) and not wt.is_segment_anomaly(None)
    assert wt.next_words_segment([dict(words=[]), ok]) is ok
    assert wt.default_alignment_heads(4, 2) == [(2, 0), (2, 1), (3, 0), (3, 1)]


def test_tokenizer_word_split_on_synthetic_vocabulary():
    from whisperlive_amd.tokenizer import Tokenizer, synthetic_tokenizer
    tk = Tokenizer(synthetic_tokenizer(4310), False)
    ids = tk.encode(" hello world, again")
    words, groups = tk.split_to_word_tokens(ids + [tk.eot])
    assert [t for g in groups for t in g] == ids + [tk.eot]                  # a partition of the tokens, in order
    assert "".join(words[:-1]) == tk.decode(ids) and len(words) == len(groups)
    assert all(w.startswith(" ") or w.strip() in ",.!?" or i == 0 for i, w in enumerate(words[:-1]))
