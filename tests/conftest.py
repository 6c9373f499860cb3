import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


# PyTorch-ROCm wheels bundle their own HIP runtime.  If libcellmap_hip.so (linked against /opt/rocm) is loaded
# first, torch later fails with "No HIP GPUs are available"; importing torch first makes both share one runtime.
try:  # pragma: no cover - depends on the environment
    import torch  # noqa: F401
except Exception:
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc
