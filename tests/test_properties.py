"""Property tests (hypothesis) of two pieces of host logic with large input spaces: the WebSocket frame codec and the
hysteresis segmentation of VAD probabilities (the size-independent properties: ranges sorted, disjoint, inside the
audio, and every window that reaches the enter threshold is covered by a range)."""
import socket
import threading

import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from whisperlive_amd import vad, ws


@settings(max_examples=60, deadline=None)
@given(payload=st.binary(min_size=0, max_size=70000), mask=st.booleans(),
       opcode=st.sampled_from([ws.OP_TEXT, ws.OP_BINARY]), split=st.integers(min_value=0, max_value=70000))
def test_frame_codec_round_trip_with_fragmentation(payload, mask, opcode, split):
    """Any payload, masked or not, whole or cut into two fragments at any point, is reassembled byte-exactly."""
    cut = min(split, len(payload))
    raw = ws.encode_frame(opcode, payload[:cut], mask=mask, fin=False) + ws.encode_frame(ws.OP_CONT, payload[cut:], mask=mask, fin=True)
    a, b = socket.socketpair()
    try:
        t = threading.Thread(target=a.sendall, args=(raw,), daemon=True)
        t.start()
        rd = ws._Reader(b)
        f1 = ws.read_frame(rd, expect_mask=mask)
        f2 = ws.read_frame(rd, expect_mask=mask)
        t.join(5)
        assert f1[:2] == (False, opcode) and f2[:2] == (True, ws.OP_CONT)
        assert f1[2] + f2[2] == payload
    finally:
        a.close(); b.close()


@settings(max_examples=80, deadline=None)
@given(data=st.binary(min_size=0, max_size=4096), key=st.binary(min_size=4, max_size=4))
def test_masking_is_an_involution(data, key):
    assert ws.apply_mask(ws.apply_mask(data, key), key) == data


@settings(max_examples=150, deadline=None)
@given(probs=st.lists(st.floats(min_value=0.0, max_value=1.0, allow_nan=False), min_size=1, max_size=400),
       tail=st.integers(min_value=1, max_value=512), thr=st.floats(min_value=0.2, max_value=0.9),
       sil=st.sampled_from([0, 100, 500, 2000]), pad=st.sampled_from([0, 30, 400]), minsp=st.sampled_from([0, 250]))
def test_speech_segments_invariants(probs, tail, thr, sil, pad, minsp):
    n = (len(probs) - 1) * 512 + tail                          # the last window may be partial
    opt = vad.VadOptions(threshold=thr, min_silence_duration_ms=sil, speech_pad_ms=pad, min_speech_duration_ms=minsp)
    segs = vad.speech_segments_from_probs(np.asarray(probs, np.float32), n, opt)
    prev_end = 0
    for s in segs:
        assert isinstance(s["start"], int) and isinstance(s["end"], int)
        assert 0 <= s["start"] < s["end"] <= n
        assert s["start"] >= prev_end                          # sorted and disjoint
        prev_end = s["end"]
    if minsp == 0:                                             # no speech run is dropped for being short
        for i, p in enumerate(probs):
            if p >= thr and i * 512 < n:
                assert any(s["start"] <= i * 512 < s["end"] for s in segs), (i, p, segs)
    # idempotent collection: concatenating the ranges gives exactly their total length, and the time map is monotone
    audio = np.arange(n, dtype=np.float32)
    chunks, _ = vad.collect_chunks(audio, segs)
    total = sum(s["end"] - s["start"] for s in segs)
    assert sum(c.shape[0] for c in chunks) == total
    if segs:
        m = vad.SpeechTimestampsMap(segs, 16000)
        ts = [m.get_original_time(t) for t in np.linspace(0, total / 16000, 7)]
        assert all(b >= a - 1e-9 for a, b in zip(ts, ts[1:]))
