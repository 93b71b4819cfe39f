"""Pick the seeds the peaked-weight GPU parity tests pin (tests/test_gpu_long_context.py, tests/test_gpu_batched_depth.py).
CPU only (the oracle): for each weight seed, run the pinned decodes on the oracle and report whether the result is unchanged
under +-amp of seeded noise on every logit (tests/helpers.py::decode_is_well_conditioned). A seed whose cases are all
well-conditioned is one where token-exactness of the GPU path is required, with no near-tie escape.
usage: PYTHONPATH=. python scripts/scan_peaked_seeds.py [small.en|large-v3] seed0 seed1 ..."""
import sys

import numpy as np

from oracle import decoding as odec
from oracle import logmel as olm
from oracle import model as omodel
from tests import helpers as H
from whisperlive_amd.specs import SPECS

name = sys.argv[1] if len(sys.argv) > 1 else "small.en"
seeds = [int(x) for x in sys.argv[2:]] or list(range(40, 48))
spec = SPECS[name]
ids = H.token_ids_for(spec.vocab)
AMP = 0.02
for seed in seeds:
    w = H.peaked_weights(spec, seed)
    o = omodel.WhisperOracle(H.oracle_spec(spec), H.f16_weights(w))
    del w
    if name == "small.en":
        pcm = olm.speech_like_pcm(30.0, seed=1234)
        cases = [("223-token prompt, 64 steps", [ids.timestamp_begin - 4] + np.random.default_rng(5).integers(0, ids.eot, size=223).tolist() + [ids.sot], 225 + 64),
                 ("[sot], 64 steps", [ids.sot], 65)]
    else:
        pcm = olm.speech_like_pcm(27.5, seed=901)      # clip 1 of tests/test_gpu_batched_depth.py
        cases = [("[sot], 64 steps", [ids.sot], 65)]
    f = olm.log_mel_spectrogram(pcm, spec.n_mels)
    enc = o.encode(olm.pad_or_trim(f[:, :-1])[None])
    for what, prompt, ml in cases:
        opts = odec.GenOptions(ids=ids, beam_size=5, patience=1.0, max_length=ml, suppress_tokens=sorted(H.default_suppress(ids)))
        ref = odec.generate(H.NetProvider(o, enc), prompt, opts)
        ok = H.decode_is_well_conditioned(o, enc, prompt, opts, ref, AMP)
        print(name, "seed", seed, what, "| tokens", len(ref.sequences_ids[0]), "distinct", len(set(ref.sequences_ids[0])),
              "score %.3f" % ref.scores[0], "| well-conditioned at +-%g:" % AMP, ok, flush=True)
