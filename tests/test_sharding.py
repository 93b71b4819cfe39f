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


def _worker_ws1(port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        clips = [np.full(10 + i, i, np.float32) for i in range(5)]
        out = sh.transcribe_clips_sharded(clips, _fake_block, rank=0, world=1, dist=dist)
        q.put([o[0] for o in out])
    finally:
        dist.destroy_process_group()


def test_world_size_one_still_runs_the_collective():
    """bench.py --rccl (tests/test_gpu_rccl.py): with an initialised process group the record exchange is a real all_gather at
    world size 1 too (until round 6 it returned early), here over gloo."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_ws1, args=(port, q))
    p.start()
    assert q.get(timeout=120) == [[i, 10 + i] for i in range(5)]
    p.join(60)
    assert p.exitcode == 0


# ---- config 5 driver: clips -> per-rank batch worker -> records -> one all_gather ------------------------------------
class _FakeBatchWorker:
    """Stands in for a rank's BatchInferenceWorker (same submit()/future contract, batches of <= max_batch_size formed
    from whatever is queued): 'transcribes' a clip into tokens derived from it."""

    def __init__(self, max_batch_size=8):
        import queue
        import threading
        self.max_batch_size = max_batch_size
        self.batches = []
        self._q = queue.Queue()
        self._t = threading.Thread(target=self._loop, daemon=True)
        self._t.start()

    def submit(self, req):
        self._q.put(req)

    def _loop(self):
        import queue
        from types import SimpleNamespace
        while True:
            batch = [self._q.get()]
            while len(batch) < self.max_batch_size:
                try:
                    batch.append(self._q.get(timeout=0.05))
                except queue.Empty:
                    break
            self.batches.append(len(batch))
            for r in batch:
                if r.audio[0] < 0:
                    r.error = RuntimeError("poisoned clip")
                else:
                    r.result = [SimpleNamespace(tokens=[int(r.audio[0]), 7], avg_logprob=-0.25, no_speech_prob=0.125),
                                SimpleNamespace(tokens=[len(r.audio)], avg_logprob=-0.5, no_speech_prob=0.5)]
                r.future.set()


def _worker5(rank, world, port, n_clips, q):
    import torch.distributed as dist
    from whisperlive_amd.batching import BatchRequest
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sh.shard_range(n_clips, rank, world)
        clips = [np.full(10 + i, i, np.float32) if lo <= i < hi else None for i in range(n_clips)]   # only the own block is materialised
        w = _FakeBatchWorker(max_batch_size=8)
        proc = sh.worker_block_processor(w, lambda c: BatchRequest(audio=c, language="en", use_vad=False), timeout_s=30)
        out = sh.transcribe_clips_sharded(clips, proc, rank=rank, world=world, dist=dist)
        q.put((rank, w.batches, [o[0] for o in out], [(o[1], o[2], o[3]) for o in out]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_config5_driver_two_ranks_gloo():
    """64 clips over 2 ranks: each rank's worker sees 4 full batches of 8 (every clip of the block is queued before the
    first result is awaited), and every rank ends with all 64 records in clip order."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker5, args=(r, 2, port, 64, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for _rank, batches, toks, floats in res:
        assert batches == [8, 8, 8, 8]
        assert toks == [[i, 7, 10 + i] for i in range(64)]
        assert floats == [(-0.25, 0.125, -0.25)] * 64


def test_config5_block_processor_surfaces_errors():
    from whisperlive_amd.batching import BatchRequest
    w = _FakeBatchWorker()
    proc = sh.worker_block_processor(w, lambda c: BatchRequest(audio=c, use_vad=False), timeout_s=10)
    with pytest.raises(RuntimeError, match="poisoned"):
        proc([np.full(4, 1, np.float32), np.full(4, -1, np.float32)])
    assert proc([]) == []


def test_bench_gpus_flag_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` by hand (no WORLD_SIZE): bench re-launches itself under torch.distributed.run with one
    rank per GPU on 127.0.0.1, forwarding its arguments; under a launcher with a different world size it refuses."""
    import subprocess
    import sys
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    assert bench.respawn_ranks(4, ["--gpus", "4", "--steps", "3"]) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert cmd[-5].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "respawn_ranks", lambda n, argv: 17)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 17
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.main()
