import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def usable_cores():
    """Host cores this process may really use: affinity mask and cgroup quota.  os.cpu_count() over-reports in containers -- the GPU box shows
    256 CPUs behind a 16-core quota, and torch's default of one intra-op thread per visible CPU then runs the CPU oracle on 256 spinning
    threads squeezed into 16 cores (round 2: 631 s for the GPU suite, 152 CPU-minutes of which most was spin-waiting)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu under gpurun)")
    import torch

    torch.set_num_threads(max(1, min(usable_cores(), 32)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
