import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def cpu_share():
    """Host cores this process may really use: the cgroup's CPU quota (a 1-GPU box of the pool shows 256 logical CPUs and a quota of 16)
    and the affinity mask.  torch sizes its pool by the logical count - 128 threads on 16 cores ran the CPU oracle 2.7x SLOWER than 16
    threads (tools/cpu_threads_probe.py, profiles/r05_cpu_threads.txt)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _default_threads():
    import torch
    return torch.get_num_threads()


DEFAULT_THREADS = _default_threads()        # torch's own choice (the logical CPU count), before pytest_configure narrows it


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(min(torch.get_num_threads(), cpu_share()))       # the oracle's thread pool: no wider than the CPU share


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
