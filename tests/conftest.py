import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cores():
    """Affinity mask capped by the cgroup CPU quota and by 16 (a GPU box advertises all 256 host cores but
    grants a share; torch-CPU oracles oversubscribed 16x run ~15x slower)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun); everything else runs on CPU")
    import torch
    torch.set_num_threads(_usable_cores())


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
