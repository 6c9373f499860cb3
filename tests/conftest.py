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


# The suite is written for ONE device: the process-wide handle of the drop-in classes takes every visible GPU by default
# (cellmapper_amd/_native.py: default_devices), which on a multi-GPU node would route every test through the multi-device
# tier.  That tier has its own tests (tests/test_gpu_multi.py lists devices explicitly); CELLMAP_TEST_ALL_GPUS=1 lifts the pin.
if not os.environ.get("CELLMAP_TEST_ALL_GPUS"):
    os.environ.setdefault("CELLMAP_DEVICES", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment(monkeypatch):
    """The library latches the CELLMAP_* knobs in cm_create; tests flip them with monkeypatch.setenv on a live handle, so
    every native call made from a test first re-reads them when the environment has changed."""
    try:
        from cellmapper_amd import _native
    except Exception:
        yield
        return
    state = {"fp": None}
    orig = _native.Handle.call

    def call(self, name, *args):
        fp = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("CELLMAP_")))
        if fp != state["fp"]:
            state["fp"] = fp
            _native.reload_knobs()
        return orig(self, name, *args)

    monkeypatch.setattr(_native.Handle, "call", call)
    yield
    if os.path.exists(_native.LIB_PATH):
        _native.reload_knobs()
