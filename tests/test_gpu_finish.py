"""GPU (-m gpu): the results of a generate are read from pinned host memory the moment the host sees the done word, while the decode step it had
already enqueued behind the finish is still running (round 6, engine.hip generate / search.hip finish_item). What must hold: a call's results are
complete and its own when it returns — whatever the previous call left in flight on the slot's stream, and however the decode ended (an
end-of-text in the first step, an end-of-text later, max_length) — for one item and for a batch whose items finish at different steps."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny(gpu):
    from whisperlive_amd.engine import HipWhisperEngine
    from whisperlive_amd.synthetic import speech_like_pcm
    from whisperlive_amd.weights import random_weights
    spec = H.TINY_EN
    eng = HipWhisperEngine(spec, random_weights(spec, seed=3))
    yield spec, eng, speech_like_pcm
    eng.close()


def _cases(spec, ids):
    eot_only = [i for i in range(spec.vocab) if i != ids.eot]
    sup = H.default_suppress(ids)
    return [
        # (prompt, options): ends on end-of-text in the first step; runs to max_length; a short one; another prompt
        ([ids.sot, ids.no_timestamps], dict(beam_size=5, max_length=40, suppress_tokens=eot_only, suppress_blank=False)),
        ([ids.sot], dict(beam_size=5, max_length=1 + 12, suppress_tokens=sup)),
        ([ids.sot], dict(beam_size=5, max_length=1 + 3, suppress_tokens=sup)),
        ([ids.sot - 1, 17, 29, 31, ids.sot], dict(beam_size=3, max_length=5 + 9, suppress_tokens=sup, num_hypotheses=2)),
    ]


def test_back_to_back_calls_return_their_own_complete_results(tiny):
    spec, eng, pcm_of = tiny
    ids = H.engine_ids(H.token_ids_for(spec.vocab))
    slot = eng.create_slot(1, 5)
    try:
        T = slot.logmel(pcm_of(20.0, seed=5))
        slot.encode(1, seek=[0], seg=[T - 1])
        cases = _cases(spec, ids)
        want = []
        for prompt, kw in cases:
            r = slot.generate([prompt], ids, **kw)[0]
            slot.timings()                                 # (waits for the stream) reference result: taken with the stream idle before and after
            r2 = slot.generate([prompt], ids, **kw)[0]
            assert r.sequences_ids == r2.sequences_ids and r.scores == r2.scores
            want.append(r)
        assert want[0].sequences_ids[0] == []              # the first case really ends on an end-of-text at once
        assert len(want[1].sequences_ids[0]) == 12
        rng = np.random.default_rng(0)
        order = rng.integers(0, len(cases), size=400)
        for n, c in enumerate(order):                      # no pause: each call starts behind whatever the previous one left in flight
            prompt, kw = cases[c]
            got = slot.generate([prompt], ids, **kw)[0]
            assert got.sequences_ids == want[c].sequences_ids, (n, c)
            assert got.scores == want[c].scores, (n, c)
            assert got.no_speech_prob == want[c].no_speech_prob, (n, c)
        tm = slot.timings()                                # waits for the tail; the device time and step count of the LAST call
        assert tm["generate_ms"] > 0.0 and tm["decode_steps"] >= 1
    finally:
        slot.close()


def test_batch_whose_items_finish_at_different_steps(tiny):
    """three items, one generate: item 1's prompt already holds max_length - 1 tokens, so it finishes in its first step; the others run
    on — the call returns when the LAST item is finished, with every item's results."""
    spec, eng, pcm_of = tiny
    ids = H.engine_ids(H.token_ids_for(spec.vocab))
    sup = H.default_suppress(ids)
    single = eng.create_slot(1, 5)
    batch = eng.create_slot(3, 5)
    try:
        clips = [pcm_of(6.0 + i, seed=40 + i) for i in range(3)]
        long_prompt = [ids.sot - 1] + [100 + i for i in range(9)] + [ids.sot]          # 11 tokens: with max_length 12 one step is left
        prompts = [[ids.sot], long_prompt, [ids.sot - 1, 55, ids.sot]]            # (one timestamp mode per call: no <|notimestamps|> mix)
        kw = dict(beam_size=5, max_length=12, suppress_tokens=sup)
        want = []
        for c, p in zip(clips, prompts):
            T = single.logmel(c); single.encode(1, seek=[0], seg=[T - 1])
            want.append(single.generate([p], ids, **kw)[0])
        assert len(want[1].sequences_ids[0]) <= 1 < len(want[0].sequences_ids[0])
        Ts = [batch.logmel(c, item=i) for i, c in enumerate(clips)]
        batch.encode(3, seek=[0] * 3, seg=[t - 1 for t in Ts])
        for n in range(60):
            got = batch.generate(prompts, ids, **kw)
            for i in range(3):
                assert got[i].sequences_ids == want[i].sequences_ids, (n, i)
                assert abs(got[i].scores[0] - want[i].scores[0]) < 1e-3, (n, i)
    finally:
        single.close(); batch.close()


def test_two_steps_per_graph_equal_one_step_per_graph(tiny):
    """One live slot on the device decodes two steps per graph launch, two or more live slots one step per launch (engine.hip graph_steps_for): the
    same calls give the same tokens, scores and step counts either way — end of text in an even step, in an odd step, odd and even step budgets
    (an odd budget ends on a one-step graph), and back to back without a pause."""
    spec, eng, pcm_of = tiny
    ids = H.engine_ids(H.token_ids_for(spec.vocab))
    sup = H.default_suppress(ids)
    eot_only = [i for i in range(spec.vocab) if i != ids.eot]
    cases = _cases(spec, ids) + [
        ([ids.sot], dict(beam_size=5, max_length=1 + n, suppress_tokens=sup)) for n in (1, 2, 7, 8)
    ] + [([ids.sot], dict(beam_size=1, max_length=1 + 9, suppress_tokens=sup)),
         ([ids.sot, ids.no_timestamps], dict(beam_size=2, max_length=30, suppress_tokens=eot_only, suppress_blank=False))]
    pcm = pcm_of(20.0, seed=5)

    def run_all(slot):
        T = slot.logmel(pcm)
        slot.encode(1, seek=[0], seg=[T - 1])
        out = []
        for rep in range(3):                                   # (three rounds back to back: every call starts behind the previous one's tail)
            for prompt, kw in cases:
                r = slot.generate([prompt], ids, **kw)[0]
                out.append((r.sequences_ids, r.scores, r.no_speech_prob))
        steps = slot.timings()["decode_steps"]
        return out, steps

    alone = eng.create_slot(1, 5)
    try:
        got2, steps2 = run_all(alone)                          # the only live slot: two steps per graph
    finally:
        alone.close()
    a = eng.create_slot(1, 5)
    b = eng.create_slot(1, 5)                                  # a second live slot: one step per graph from here on
    try:
        got1, steps1 = run_all(a)
    finally:
        a.close(); b.close()
    assert got2 == got1
    assert steps2 == steps1
