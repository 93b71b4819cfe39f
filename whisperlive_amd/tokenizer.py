"""Whisper tokenizer glue over a Hugging Face `tokenizers.Tokenizer` — the role faster_whisper.tokenizer.Tokenizer plays
for the reference (constructed at whisper_live/transcriber/transcriber_faster_whisper.py:909-914 and
whisper_live/batch_inference.py:293-298; attributes used at :981-1001, :1144-1145, :1491-1510, :1837, :1845-1849).
Every special id is resolved BY NAME from the vocabulary, never hard-coded.

`synthetic_tokenizer()` builds a tokenizer.json-equivalent with the exact Whisper special-token layout around a tiny
byte-level vocabulary: no real vocabulary exists offline, and the end-to-end tests / bench need one."""
from __future__ import annotations

from functools import cached_property
from typing import List, Optional, Sequence, Tuple

LANGUAGE_CODES: Tuple[str, ...] = (
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id", "hi", "fi",
    "vi", "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg", "lt", "la", "mi", "ml",
    "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br", "eu", "is", "hy", "ne", "mn", "bs",
    "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so", "af", "oc", "ka", "be", "tg", "sd", "gu", "am",
    "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt", "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln",
    "ha", "ba", "jw", "su", "yue",
)
TASKS = ("transcribe", "translate")


class Tokenizer:
    def __init__(self, tokenizer, multilingual: bool, task: Optional[str] = None, language: Optional[str] = None):
        self.tokenizer = tokenizer
        if multilingual:
            if task not in TASKS:
                raise ValueError(f"'{task}' is not a valid task (accepted tasks: {', '.join(TASKS)})")
            if language not in LANGUAGE_CODES:
                raise ValueError(f"'{language}' is not a valid language code")
            self.task = self._id(f"<|{task}|>")
            self.language = self._id(f"<|{language}|>")
            self.language_code = language
        else:
            self.task = None
            self.language = None
            self.language_code = "en"

    def _id(self, token: str, required: bool = True) -> Optional[int]:
        i = self.tokenizer.token_to_id(token)
        if i is None and required:
            raise KeyError(f"special token {token} is not in the vocabulary")
        return i

    @cached_property
    def transcribe(self) -> int:
        return self._id("<|transcribe|>")

    @cached_property
    def translate(self) -> int:
        return self._id("<|translate|>")

    @cached_property
    def sot(self) -> int:
        return self._id("<|startoftranscript|>")

    @cached_property
    def sot_lm(self) -> int:
        return self._id("<|startoflm|>")

    @cached_property
    def sot_prev(self) -> int:
        return self._id("<|startofprev|>")

    @cached_property
    def eot(self) -> int:
        return self._id("<|endoftext|>")

    @cached_property
    def no_timestamps(self) -> int:
        return self._id("<|notimestamps|>")

    @cached_property
    def no_speech(self) -> int:
        i = self._id("<|nospeech|>", required=False)
        return i if i is not None else self._id("<|nocaptions|>")

    @property
    def timestamp_begin(self) -> int:
        return self.no_timestamps + 1

    @property
    def sot_sequence(self) -> List[int]:
        seq = [self.sot]
        if self.language is not None:
            seq.append(self.language)
        if self.task is not None:
            seq.append(self.task)
        return seq

    @cached_property
    def blank(self) -> int:
        ids = self.encode(" ")
        return ids[0] if ids else -1

    def encode(self, text: str) -> List[int]:
        return self.tokenizer.encode(text, add_special_tokens=False).ids

    def decode(self, tokens: Sequence[int]) -> str:
        return self.tokenizer.decode([t for t in tokens if t < self.eot])

    def decode_with_timestamps(self, tokens: Sequence[int]) -> str:
        out, run = [], []
        for t in tokens:
            if t >= self.timestamp_begin:
                if run:
                    out.append(self.tokenizer.decode(run))
                    run = []
                out.append(f"<|{(t - self.timestamp_begin) * 0.02:.2f}|>")
            elif t < self.eot:
                run.append(t)
        if run:
            out.append(self.tokenizer.decode(run))
        return "".join(out)

    # ---- word boundaries for word-level timestamps (faster_whisper.tokenizer.Tokenizer.split_to_word_tokens, used at
    # transcriber_faster_whisper.py:1670; un-vendored — the published openai/whisper tokenizer.py rules)
    def split_to_word_tokens(self, tokens: Sequence[int]) -> Tuple[List[str], List[List[int]]]:
        if self.language_code in ("zh", "ja", "th", "lo", "my", "yue"):
            # scripts written without spaces: every complete unicode character sequence is its own "word"
            return self.split_tokens_on_unicode(tokens)
        return self.split_tokens_on_spaces(tokens)

    def split_tokens_on_unicode(self, tokens: Sequence[int]) -> Tuple[List[str], List[List[int]]]:
        """cut wherever the tokens decoded so far form valid text (no dangling byte-level fragment, U+FFFD)"""
        full = self.decode_with_timestamps(tokens)
        bad = "\ufffd"
        words: List[str] = []
        groups: List[List[int]] = []
        cur: List[int] = []
        offset = 0
        for t in tokens:
            cur.append(t)
            text = self.decode_with_timestamps(cur)
            at = text.find(bad)
            # a replacement character is only genuine if the full decoding has one at the same place
            if at < 0 or (offset + at < len(full) and full[offset + at] == bad):
                words.append(text)
                groups.append(cur)
                cur = []
                offset += len(text)
        return words, groups

    def split_tokens_on_spaces(self, tokens: Sequence[int]) -> Tuple[List[str], List[List[int]]]:
        """unicode pieces merged into words: a piece opens a new word if it is a special token, starts with a space or
        is bare punctuation; anything else continues the previous word"""
        import string
        pieces, piece_tokens = self.split_tokens_on_unicode(tokens)
        words: List[str] = []
        groups: List[List[int]] = []
        for piece, toks in zip(pieces, piece_tokens):
            opens = toks[0] >= self.eot or piece.startswith(" ") or piece.strip() in string.punctuation or not words
            if opens:
                words.append(piece)
                groups.append(list(toks))
            else:
                words[-1] += piece
                groups[-1].extend(toks)
        return words, groups

    @cached_property
    def non_speech_tokens(self) -> Tuple[int, ...]:
        """Ids of symbol / bracket / music-note tokens that Whisper suppresses by default (the published OpenAI
        list, as faster-whisper evaluates it): single-token encodings of each symbol with and without a leading
        space, plus the first token of ' -' and " '"; music notes contribute their first token even when multi-token.
        A function of the underlying tokenizer alone, and `transcribe` builds a fresh `Tokenizer` wrapper per call (as the reference
        does, transcriber_faster_whisper.py:880-885): memoised per tokenizer object — the 110 encodes were 0.6 ms of every chunk's
        ~1 ms of host time (round 6)."""
        hit = _NON_SPEECH_CACHE.get(id(self.tokenizer))
        if hit is not None and hit[0] is self.tokenizer:
            return hit[1]
        out = self._non_speech_tokens()
        if len(_NON_SPEECH_CACHE) >= 32:
            _NON_SPEECH_CACHE.clear()
        _NON_SPEECH_CACHE[id(self.tokenizer)] = (self.tokenizer, out)          # (the object is kept: its id cannot be recycled under us)
        return out

    def _non_speech_tokens(self) -> Tuple[int, ...]:
        symbols = list("\"#()*+/:;<=>@[\\]^_`{|}~「」『』")
        symbols += "<< >> <<< >>> -- --- -( -[ (' (\" (( )) ((( ))) [[ ]] {{ }} ♪♪ ♪♪♪".split()
        music = set("♩♪♫♬♭♮♯")
        found = set()
        for lead in (" -", " '"):
            ids = self.encode(lead)
            if ids:
                found.add(ids[0])
        for sym in symbols + sorted(music):
            for ids in (self.encode(sym), self.encode(" " + sym)):
                if ids and (len(ids) == 1 or sym in music):
                    found.add(ids[0])
        return tuple(sorted(found))

    def language_token_ids(self) -> List[Tuple[str, int]]:
        """(code, id) of every language token present in the vocabulary, in vocabulary order."""
        out = []
        for code in LANGUAGE_CODES:
            i = self.tokenizer.token_to_id(f"<|{code}|>")
            if i is not None:
                out.append((code, i))
        return sorted(out, key=lambda x: x[1])


_NON_SPEECH_CACHE: dict = {}


def synthetic_tokenizer(vocab_size: int, n_languages: Optional[int] = None, timestamps: bool = True):
    """A `tokenizers.Tokenizer` with Whisper's special-token layout for a vocabulary of `vocab_size` ids:
    [0, eot): byte-level pieces (256 byte tokens + filler words), then <|endoftext|>, <|startoftranscript|>,
    99 (100 for large-v3) language tokens — English-only vocabularies carry them too —, <|translate|>, <|transcribe|>,
    <|startoflm|>, <|startofprev|>, <|nospeech|>, <|notimestamps|>, <|0.00|> ... <|30.00|>.
    For 51864 / 51865 / 51866 this reproduces the real id layout."""
    from tokenizers import Tokenizer as HFTokenizer
    from tokenizers import decoders, models, pre_tokenizers

    n_lang = n_languages if n_languages is not None else (100 if vocab_size == 51866 else 99)
    n_special = 1 + 1 + n_lang + 6 + 1501
    n_text = vocab_size - n_special
    if n_text < 256:
        raise ValueError("vocab_size too small for the Whisper special-token layout")
    byte_alphabet = pre_tokenizers.ByteLevel.alphabet()
    vocab = {ch: i for i, ch in enumerate(sorted(byte_alphabet))}
    i = len(vocab)
    while i < n_text:
        vocab[f"Ġw{i}"] = i
        i += 1
    tok = HFTokenizer(models.BPE(vocab=vocab, merges=[]))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    specials = ["<|endoftext|>", "<|startoftranscript|>"]
    specials += [f"<|{c}|>" for c in LANGUAGE_CODES[:n_lang]]
    specials += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
    if timestamps:
        specials += [f"<|{k * 0.02:.2f}|>" for k in range(1501)]
    # timestamps=False: the tokenizer.json of older converted checkpoints, which ends at <|notimestamps|>
    tok.add_special_tokens(specials)
    assert tok.get_vocab_size() == vocab_size - (0 if timestamps else 1501), (tok.get_vocab_size(), vocab_size)
    return tok
