import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def usable_cpus(cap: int = 16) -> int:
    """Host threads this process may really use: the affinity mask and the cgroup CPU quota, not os.cpu_count(). A container
    on a 256-thread host reports 256 and is scheduled on a fraction of them; torch with 256 threads then crawls (the
    large-v3 oracle decode took 3.5 s per step on the GPU box, profiles/r3b_pytest_durations.txt, against 0.15 s here)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # hermetic: artefact look-ups (whisperlive_amd/artifacts.py) may read caches but never start a download from inside the test suite
    os.environ.setdefault("WLX_NO_DOWNLOAD", "1")
    try:                                   # the CPU oracle (torch fp32) on the cores this container really has
        import torch
        torch.set_num_threads(usable_cpus())
    except Exception:  # noqa: BLE001
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _has_gpu():
        pytest.skip("no GPU visible")
    return True
