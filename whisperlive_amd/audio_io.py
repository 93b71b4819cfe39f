"""Audio file input for the transcription path: WAV (PCM 8/16/24/32-bit, float32) and FLAC -> mono float32 at 16 kHz.

The reference accepts a path or file object in ``WhisperModel.transcribe`` and decodes it with PyAV/FFmpeg
(``faster_whisper.audio.decode_audio``, called at whisper_live/transcriber/transcriber_faster_whisper.py:821); neither is
available offline, and the only audio the reference ships — ``assets/jfk.flac``, the clip its end-to-end test transcribes
(tests/test_server.py:73-118) — is 24-bit stereo 44.1 kHz FLAC. This module is a dependency-free reader for exactly that
job: a FLAC decoder (fixed / LPC / verbatim / constant subframes, partitioned Rice residuals, stereo decorrelation; checked
against the MD5 of the unencoded audio that every FLAC file carries in its STREAMINFO block), a RIFF/WAVE reader, channel
down-mix and polyphase resampling (``scipy.signal.resample_poly``). It is host-side plumbing: FFmpeg's resampler is a
different filter, so samples agree with the reference's decode to resampler tolerance, not bit for bit.
"""
from __future__ import annotations

import hashlib
import struct
from math import gcd
from typing import BinaryIO, Tuple, Union

import numpy as np

PathOrFile = Union[str, bytes, BinaryIO]


def _read_all(src: PathOrFile) -> bytes:
    if isinstance(src, (bytes, bytearray)):
        return bytes(src)
    if isinstance(src, str):
        with open(src, "rb") as f:
            return f.read()
    return src.read()


# ---------------------------------------------------------------------------------------------------------------- WAV
def read_wav(src: PathOrFile) -> Tuple[np.ndarray, int]:
    """RIFF/WAVE -> (float32 [frames, channels] in [-1, 1), sample rate). PCM 8/16/24/32-bit, IEEE float 32/64,
    WAVE_FORMAT_EXTENSIBLE wrappers of those."""
    b = _read_all(src)
    if b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        cid, size = b[pos:pos + 4], struct.unpack_from("<I", b, pos + 4)[0]
        body = b[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            data = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("WAVE file without fmt / data chunk")
    tag, ch, sr, _br, _ba, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if tag == 0xFFFE and len(fmt) >= 26:                      # extensible: the real format is the first GUID word
        tag = struct.unpack_from("<H", fmt, 24)[0]
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(data, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(data[: len(data) // 2 * 2], "<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            raw = np.frombuffer(data[: len(data) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
            v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
            x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
        elif bits == 32:
            x = np.frombuffer(data[: len(data) // 4 * 4], "<i4").astype(np.float32) / 2147483648.0
        else:
            raise ValueError(f"unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(data, "<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"unsupported WAVE format tag {tag}")
    return x[: x.size // ch * ch].reshape(-1, ch), sr


# ---------------------------------------------------------------------------------------------------------------- FLAC
class _Bits:
    """MSB-first bit reader over a bytes object (one FLAC frame at a time is small enough for a Python int)."""

    def __init__(self, data: bytes, pos: int = 0):
        self.d, self.p = data, pos * 8

    def read(self, n: int) -> int:
        if n == 0:
            return 0
        p, e = self.p, self.p + n
        b0, b1 = p >> 3, (e + 7) >> 3
        v = int.from_bytes(self.d[b0:b1], "big")
        self.p = e
        return (v >> (b1 * 8 - e)) & ((1 << n) - 1)

    def read_signed(self, n: int) -> int:
        v = self.read(n)
        return v - (1 << n) if v >> (n - 1) else v

    def unary(self) -> int:                 # zeros before the next 1 bit
        n = 0
        while True:
            byte_i, off = self.p >> 3, self.p & 7
            chunk = self.d[byte_i] & (0xFF >> off)
            if chunk:
                lead = 8 - off - chunk.bit_length()
                self.p += lead + 1
                return n + lead
            n += 8 - off
            self.p += 8 - off

    def align(self):
        self.p = (self.p + 7) & ~7

    @property
    def byte_pos(self) -> int:
        return self.p >> 3


_BLOCK = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608}
_FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


def _residual(bits: _Bits, data: bytes, n: int, order: int) -> np.ndarray:
    """Partitioned Rice coding. The unary / binary split of every symbol is found with a table of 'next set bit' and a
    table of k-bit windows built with numpy over the partition's bit range; the per-symbol Python loop only indexes them."""
    method = bits.read(2)
    if method > 1:
        raise ValueError("reserved residual coding method")
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    porder = bits.read(4)
    nparts = 1 << porder
    out = np.empty(n - order, np.int64)
    o = 0
    for part in range(nparts):
        cnt = (n >> porder) - (order if part == 0 else 0) if porder else n - order
        k = bits.read(pbits)
        if k == esc:
            w = bits.read(5)
            for i in range(cnt):
                out[o + i] = bits.read_signed(w) if w else 0
            o += cnt
            continue
        if cnt == 0:
            continue
        # bit array from the current position: a generous bound, then grow if the partition is longer
        start = bits.p
        span = cnt * (k + 12) + 64
        while True:
            b0, b1 = start >> 3, min(len(data), (start + span + 7) >> 3)
            arr = np.unpackbits(np.frombuffer(data, np.uint8, b1 - b0, b0))
            arr = arr[start & 7:]
            ones = np.flatnonzero(arr)
            vals = None
            if k:
                pad = np.concatenate([arr, np.zeros(k, np.uint8)]).astype(np.int64)
                vals = np.zeros(arr.size, np.int64)
                for i in range(k):
                    vals = (vals << 1) | pad[1 + i: 1 + i + arr.size]       # k bits FOLLOWING position j (the stop bit)
            nxt = np.searchsorted(ones, np.arange(arr.size))                # index into `ones` of the next set bit at or after i
            pos, ok = 0, True
            res = []
            on, nn, size = ones.tolist(), nxt.tolist(), arr.size
            vl = vals.tolist() if vals is not None else None
            for _ in range(cnt):
                if pos >= size or nn[pos] >= len(on):
                    ok = False
                    break
                j = on[nn[pos]]
                if j + k >= size and b1 < len(data):
                    ok = False
                    break
                u = ((j - pos) << k) | (vl[j] if k else 0)
                res.append((u >> 1) ^ -(u & 1))
                pos = j + 1 + k
            if ok:
                break
            if b1 >= len(data):
                raise ValueError("FLAC residual runs past the end of the data")
            span *= 2
        out[o:o + cnt] = res
        o += cnt
        bits.p = start + pos
    return out


def _subframe(bits: _Bits, data: bytes, n: int, bps: int) -> np.ndarray:
    if bits.read(1):
        raise ValueError("FLAC subframe padding bit set")
    typ = bits.read(6)
    wasted = 0
    if bits.read(1):
        wasted = bits.unary() + 1
        bps -= wasted
    if typ == 0:
        x = np.full(n, bits.read_signed(bps), np.int64)
    elif typ == 1:
        x = np.array([bits.read_signed(bps) for _ in range(n)], np.int64)
    elif 8 <= typ <= 12 or typ >= 32:
        if typ >= 32:
            order = (typ & 31) + 1
            warm = [bits.read_signed(bps) for _ in range(order)]
            prec = bits.read(4) + 1
            shift = bits.read_signed(5)
            coefs = [bits.read_signed(prec) for _ in range(order)]
        else:
            order = typ - 8
            warm = [bits.read_signed(bps) for _ in range(order)]
            coefs, shift = list(_FIXED[order]), 0
        res = _residual(bits, data, n, order).tolist()
        xs = warm + [0] * (n - order)
        if order == 0:
            xs = res
        else:
            rc = coefs[::-1]                                   # oldest sample first, to zip against xs[i-order:i]
            for i in range(order, n):
                acc = 0
                for c, v in zip(rc, xs[i - order:i]):
                    acc += c * v
                xs[i] = res[i - order] + (acc >> shift)
        x = np.asarray(xs, np.int64)
    else:
        raise ValueError(f"reserved FLAC subframe type {typ}")
    return x << wasted if wasted else x


def read_flac(src: PathOrFile, verify_md5: bool = True) -> Tuple[np.ndarray, int]:
    """FLAC -> (float32 [frames, channels] in [-1, 1), sample rate). Verifies the decoded samples against the MD5 signature
    in STREAMINFO unless told not to (a mismatch raises: a FLAC decoder that is wrong is wrong loudly)."""
    b = _read_all(src)
    if b[:4] != b"fLaC":
        raise ValueError("not a FLAC file")
    pos, info = 4, None
    while True:
        last, typ = b[pos] >> 7, b[pos] & 0x7F
        ln = int.from_bytes(b[pos + 1:pos + 4], "big")
        if typ == 0:
            info = b[pos + 4: pos + 4 + ln]
        pos += 4 + ln
        if last:
            break
    if info is None:
        raise ValueError("FLAC file without STREAMINFO")
    x = int.from_bytes(info[10:18], "big")
    sr, ch, bps, total = x >> 44, ((x >> 41) & 7) + 1, ((x >> 36) & 31) + 1, x & ((1 << 36) - 1)
    md5 = info[18:34]
    chans = [[] for _ in range(ch)]
    done = 0
    while pos < len(b) - 2 and (total == 0 or done < total):
        if b[pos] != 0xFF or (b[pos + 1] & 0xFE) != 0xF8:
            raise ValueError(f"lost FLAC frame sync at byte {pos}")
        bits = _Bits(b, pos)
        bits.read(15)
        bits.read(1)                                                     # blocking strategy
        bs_code, sr_code, ch_code, ss_code = bits.read(4), bits.read(4), bits.read(4), bits.read(3)
        bits.read(1)
        first = bits.read(8)                                             # UTF-8-style coded frame / sample number
        extra = 0 if first < 0x80 else (first ^ 0xFF).bit_length() and (8 - (first ^ 0xFF).bit_length() - 1)
        for _ in range(extra):
            bits.read(8)
        if bs_code == 6:
            n = bits.read(8) + 1
        elif bs_code == 7:
            n = bits.read(16) + 1
        elif bs_code in _BLOCK:
            n = _BLOCK[bs_code]
        elif bs_code >= 8:
            n = 256 << (bs_code - 8)
        else:
            raise ValueError("reserved FLAC block size")
        if sr_code == 12:
            bits.read(8)
        elif sr_code in (13, 14):
            bits.read(16)
        fb = {0: bps, 1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}.get(ss_code)
        if fb is None:
            raise ValueError("reserved FLAC sample size")
        bits.read(8)                                                     # header CRC-8
        if ch_code < 8:
            subs = [_subframe(bits, b, n, fb) for _ in range(ch_code + 1)]
        elif ch_code == 8:                                               # left, side
            l, s = _subframe(bits, b, n, fb), _subframe(bits, b, n, fb + 1)
            subs = [l, l - s]
        elif ch_code == 9:                                               # side, right
            s, r = _subframe(bits, b, n, fb + 1), _subframe(bits, b, n, fb)
            subs = [s + r, r]
        elif ch_code == 10:                                              # mid, side
            m, s = _subframe(bits, b, n, fb), _subframe(bits, b, n, fb + 1)
            m = (m << 1) | (s & 1)
            subs = [(m + s) >> 1, (m - s) >> 1]
        else:
            raise ValueError("reserved FLAC channel assignment")
        bits.align()
        bits.read(16)                                                    # frame CRC-16
        pos = bits.byte_pos
        for c in range(ch):
            chans[c].append(subs[c])
        done += n
    pcm = np.stack([np.concatenate(c) for c in chans], axis=1)
    if total:
        pcm = pcm[:total]
    if verify_md5 and md5 != bytes(16):
        nb = (bps + 7) // 8
        raw = pcm.astype("<i8").view(np.uint8).reshape(pcm.shape[0], ch, 8)[:, :, :nb]
        if hashlib.md5(raw.tobytes()).digest() != md5:
            raise ValueError("FLAC decode does not match the file's STREAMINFO MD5 signature")
    return (pcm / float(1 << (bps - 1))).astype(np.float32), sr


# ---------------------------------------------------------------------------------------------------------------- API
def load_audio(src: PathOrFile, sampling_rate: int = 16000) -> np.ndarray:
    """File / bytes / file object -> mono float32 waveform at `sampling_rate` (the shape ``transcribe`` expects)."""
    b = _read_all(src)
    if b[:4] == b"fLaC":
        x, sr = read_flac(b)
    elif b[:4] == b"RIFF":
        x, sr = read_wav(b)
    else:
        raise ValueError("unsupported audio container (WAV and FLAC are read natively; decode other formats to 16 kHz "
                         "float32 PCM first)")
    mono = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]
    if sr != sampling_rate:
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(sampling_rate))
        mono = resample_poly(mono.astype(np.float64), sampling_rate // g, sr // g)
    return np.ascontiguousarray(mono, dtype=np.float32)
