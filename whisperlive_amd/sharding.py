"""Multi-GPU layout of the hot path: client streams are independent units, so they SHARD — one
process per GPU (torch.distributed rank = GPU), one engine (weights replica) + a slot pool per rank, stream i on GPU
``i mod n``; there is no collective on the streaming path. The reference's own multi-GPU story is replica selection
only (ROCm_whisper.md:66 ``HIP_VISIBLE_DEVICES``; CT2 ``device_index=[...]``,
whisper_live/transcriber/transcriber_faster_whisper.py:579,598-601).

Batched mode (whisper_live/batch_inference.py; config 5: N pre-recorded clips over all GPUs) has exactly one
exchange step: every rank transcribes a contiguous block of the clips, then the results (<= 448 token ids + 3 floats
per clip, a fixed 2 KiB record) are all-gathered — RCCL over xGMI on the GPU box (backend "nccl"), gloo on CPU in the
tests. Message size makes link bandwidth irrelevant; it is a completion barrier plus a few KiB.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

RECORD_INTS = 512          # 2 KiB: [n_tokens, score_bits, no_speech_bits, avg_logprob_bits, tokens[<=448], pad]
MAX_TOKENS = 448


def assign_gpu(stream_index: int, n_gpus: int) -> int:
    if n_gpus < 1:
        raise ValueError("n_gpus must be >= 1")
    return stream_index % n_gpus


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one item, earlier ranks get the extras."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_record(tokens: Sequence[int], score: float, no_speech_prob: float, avg_logprob: float = 0.0) -> np.ndarray:
    rec = np.zeros(RECORD_INTS, dtype=np.int32)
    n = min(len(tokens), MAX_TOKENS)
    rec[0] = n
    rec[1:4] = np.asarray([score, no_speech_prob, avg_logprob], dtype=np.float32).view(np.int32)
    rec[4:4 + n] = np.asarray(tokens[:n], dtype=np.int32)
    return rec


def unpack_record(rec: np.ndarray) -> Tuple[List[int], float, float, float]:
    n = int(rec[0])
    score, nsp, alp = np.asarray(rec[1:4], dtype=np.int32).view(np.float32).tolist()
    return [int(t) for t in rec[4:4 + n]], score, nsp, alp


def all_gather_records(local: np.ndarray, n_total: int, rank: int, world: int, dist=None, device: str = "cpu") -> np.ndarray:
    """local: int32 [hi-lo, RECORD_INTS] for this rank's block -> int32 [n_total, RECORD_INTS] on every rank.
    Blocks are padded to the largest block so ONE fixed-size all_gather suffices. With an initialised process group the
    collective runs at world size 1 too (bench.py --rccl: the RCCL path executed on a one-GPU box); `dist=None` skips it."""
    if dist is None:
        return local.reshape(n_total, RECORD_INTS)
    import torch
    per = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    buf = torch.zeros((per, RECORD_INTS), dtype=torch.int32, device=device)
    if local.size:
        buf[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(outs[r][: hi - lo].cpu().numpy())
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, RECORD_INTS), np.int32)


def transcribe_clips_sharded(clips: Sequence[np.ndarray], process_block: Callable[[Sequence[np.ndarray]], List[np.ndarray]],
                             rank: int = 0, world: int = 1, dist=None, device: str = "cpu") -> List[Tuple[List[int], float, float, float]]:
    """Config-5 driver: rank r runs `process_block` (clips -> one record per clip, e.g. through a
    BatchInferenceWorker on its GPU) on its contiguous block, then all ranks exchange the records."""
    lo, hi = shard_range(len(clips), rank, world)
    recs = process_block(clips[lo:hi]) if hi > lo else []
    local = np.stack(recs).astype(np.int32) if recs else np.zeros((0, RECORD_INTS), np.int32)
    if local.shape[0] != hi - lo:
        raise RuntimeError("process_block must return one record per clip")
    gathered = all_gather_records(local, len(clips), rank, world, dist, device)
    return [unpack_record(r) for r in gathered]


def worker_block_processor(worker, make_request: Callable[[np.ndarray], "object"], timeout_s: float = 600.0
                           ) -> Callable[[Sequence[np.ndarray]], List[np.ndarray]]:
    """`process_block` for transcribe_clips_sharded on top of a rank's BatchInferenceWorker — what the reference's
    server does with N pre-recorded clips in batched mode (whisper_live/server.py:665-673 starts the worker,
    whisper_live/batch_inference.py:155-187 collects up to max_batch_size requests per batch): EVERY clip of the block is
    submitted before the first result is awaited, so the worker forms full batches; each result becomes one 2 KiB record
    (all generated tokens of the clip, avg_logprob in both score fields, no_speech_prob)."""
    def process(clips: Sequence[np.ndarray]) -> List[np.ndarray]:
        reqs = [make_request(c) for c in clips]
        for r in reqs:
            worker.submit(r)
        recs = []
        for i, r in enumerate(reqs):
            if not r.future.wait(timeout_s):
                raise TimeoutError(f"clip {i} of the block was not transcribed within {timeout_s} s")
            if r.error is not None:
                raise r.error
            segs = list(r.result or [])
            toks = [int(t) for sg in segs for t in sg.tokens]
            alp = float(segs[0].avg_logprob) if segs else 0.0
            nsp = float(segs[0].no_speech_prob) if segs else 1.0
            recs.append(pack_record(toks, alp, nsp, alp))
        return recs
    return process
