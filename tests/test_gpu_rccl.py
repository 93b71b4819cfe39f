"""RCCL executed on hardware (VERDICT r05 "next round" 4b): the build pool has one-GPU boxes only, so `backend="nccl"` never ran in
rounds 1-5 — the N-rank path was covered by world-size-2 gloo tests (tests/test_sharding.py) and a gloo rehearsal. Here the SAME code
runs through a world-size-1 RCCL process group with DEVICE tensors:

* whisperlive_amd.sharding.all_gather_records (the one data-path collective of the batched mode) + the barrier and
  the max-over-ranks all_reduce bench.py brackets its timed region with;
* `bench.py --config 5 --gpus 1 --rccl` (configs[4]'s driver: worker -> records -> all_gather) and the headline driver with --rccl,
  at small shapes.

Each case is a subprocess with a timeout: a collective that hangs fails the test instead of the run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), WLX_QUIET="1")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


_COLLECTIVES = r"""
import numpy as np, torch, torch.distributed as dist
from whisperlive_amd import sharding as sh
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
recs = np.stack([sh.pack_record(list(range(100 + i, 100 + i + 3 * i)), -0.25 * i, 0.01 * i, -0.5 * i) for i in range(7)])
out = sh.all_gather_records(recs, 7, 0, 1, dist, device="cuda:0")
assert out.shape == (7, sh.RECORD_INTS) and np.array_equal(out, recs)
assert sh.unpack_record(out[3])[0] == list(range(103, 112))
got = sh.transcribe_clips_sharded([np.zeros(4, np.float32)] * 5, lambda clips: [sh.pack_record([1, 2, 3], -1.0, 0.5, -1.0) for _ in clips],
                                  rank=0, world=1, dist=dist, device="cuda:0")
assert len(got) == 5 and got[4][0] == [1, 2, 3] and abs(got[4][2] - 0.5) < 1e-7
t = torch.tensor([3.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
lt = torch.arange(4, dtype=torch.float64, device="cuda:0")
g = [torch.zeros_like(lt)]
dist.all_gather(g, lt)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 3.5 and torch.equal(g[0], lt)
dist.destroy_process_group()
print("RCCL_OK", torch.version.hip)
"""


def test_record_all_gather_and_timing_collectives_through_rccl(gpu):
    r = subprocess.run([sys.executable, "-c", _COLLECTIVES], cwd=ROOT, env=_env(29611), capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _bench(args, port, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=_env(port), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_config5_driver_through_rccl(gpu):
    out = _bench(["--config", "5", "--gpus", "1", "--rccl", "--model", "tiny.en", "--clips", "6", "--max-batch", "4", "--lanes", "1",
                  "--steps", "1", "--warmup", "1", "--decode-steps", "6", "--no-pmc"], 29613)
    assert out["n_gpus"] == 1 and out["collectives"] == {"backend": "nccl (RCCL, device tensors)", "world_size": 1}
    assert out["value"] > 0 and out["config"]["clips"] == 6 and out["tokens_per_clip"]["min"] == out["tokens_per_clip"]["max"] > 0
    assert "rehearsal" not in out


def test_headline_driver_through_rccl(gpu):
    out = _bench(["--gpus", "1", "--rccl", "--model", "tiny.en", "--steps", "2", "--warmup", "1", "--decode-steps", "8",
                  "--no-cpu-baseline", "--no-stream", "--no-throughput", "--no-pmc"], 29615)
    assert out["n_gpus"] == 1 and out["collectives"]["backend"].startswith("nccl") and out["value"] > 0 and out["steps"] == 2
