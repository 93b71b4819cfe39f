"""CPU tests of the multi-GPU layout: stream -> GPU assignment, contiguous clip blocks, the fixed
2 KiB result record, and the ONE exchange step of batched mode run for real over torch.distributed with the gloo
backend at world_size 2 (the GPU box runs the same code over RCCL/xGMI with backend "nccl")."""
import os
import socket

import numpy as np
import pytest

from whisperlive_amd import sharding as sh


def test_assignment_and_blocks():
    assert [sh.assign_gpu(i, 8) for i in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert [sh.shard_range(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]
    blocks = [sh.shard_range(10, r, 4) for r in range(4)]
    assert blocks == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sh.shard_range(2, 3, 4) == (2, 2)
    with pytest.raises(ValueError):
        sh.shard_range(4, 4, 4)


def test_record_round_trip():
    toks = list(range(50363, 50363 + 448))
    rec = sh.pack_record(toks + [1, 2, 3], -0.3125, 0.0625, -0.5)
    assert rec.shape == (512,) and rec.dtype == np.int32
    t, s, n, a = sh.unpack_record(rec)
    assert t == toks and (s, n, a) == (-0.3125, 0.0625, -0.5)
    assert sh.unpack_record(sh.pack_record([], 0.0, 1.0))[0] == []


def _fake_block(clips):
    """Stand-in for a per-GPU BatchInferenceWorker: 'tokens' derived from the clip so the gather can be checked."""
    return [sh.pack_record([int(c[0]), len(c)], float(c[0]) * 0.5, 0.25) for c in clips]


def test_single_rank_path():
    clips = [np.full(10 + i, i, np.float32) for i in range(5)]
    out = sh.transcribe_clips_sharded(clips, _fake_block)
    assert [o[0] for o in out] == [[i, 10 + i] for i in range(5)]


def _worker(rank, world, port, n_clips, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clips = [np.full(10 + i, i, np.float32) for i in range(n_clips)]
        seen = []

        def block(cl):
            seen.extend(int(c[0]) for c in cl)
            return _fake_block(cl)

        out = sh.transcribe_clips_sharded(clips, block, rank=rank, world=world, dist=dist)
        q.put((rank, seen, [o[0] for o in out], [o[1] for o in out]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [7, 1])
def test_two_rank_gloo_gather(n_clips):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    lo0, hi0 = sh.shard_range(n_clips, 0, 2)
    assert res[0][1] == list(range(lo0, hi0)) and res[1][1] == list(range(hi0, n_clips))      # disjoint blocks
    for _rank, _seen, toks, scores in res:                                                     # every rank has everything
        assert toks == [[i, 10 + i] for i in range(n_clips)] and scores == [i * 0.5 for i in range(n_clips)]
