"""Backend gate for the k-NN methods.

The reference's only "plugin" mechanism is a ``method`` string checked against optional imports
(``check.py:39-89``: ``Checker.check`` / ``check_deps``; ``knn.py:376-465``).  This module mirrors it: backends
are registered by name, and ``check_deps("hip")`` raises ``RuntimeError`` with a build hint when
``libcellmap_hip.so`` cannot be loaded or no GPU is visible -- there is no silent CPU fallback.
"""
from __future__ import annotations

import importlib
from typing import Callable


class Checker:
    """Availability check for one backend (same role as the reference's ``Checker``)."""

    def __init__(self, name: str, probe: Callable[[], None] | None = None, install_hint: str | None = None) -> None:
        self.name = name
        self.probe = probe
        self.install_hint = install_hint
        self.vreq_hint = f"This function relies on {self.name}."

    def check(self) -> None:
        try:
            if self.probe is not None:
                self.probe()
            else:
                importlib.import_module(self.name)
        except (ModuleNotFoundError, RuntimeError, OSError) as e:
            raise RuntimeError(" ".join(filter(None, [self.vreq_hint, str(e), self.install_hint]))) from e


def _probe_hip() -> None:
    from . import _native

    _native.default_handle()  # loads the .so and creates a context on the GPU; raises RuntimeError otherwise


CHECKERS = {
    "hip": Checker("libcellmap_hip (MI355X backend)", probe=_probe_hip),
}

#: methods accepted by ``Neighbors.compute_neighbors`` / ``CellMapper.map(knn_method=...)``.  All of them run on
#: the HIP backend here; the reference's names are accepted so existing call sites keep working unchanged.
KNN_METHODS = ("hip", "sklearn", "faiss", "rapids", "pynndescent")


def check_deps(*args) -> None:
    """Check that the named backends are usable (raises ``RuntimeError`` otherwise)."""
    for item in args:
        if item not in CHECKERS:
            raise RuntimeError(f"Dependency '{item}' is not registered in CHECKERS.")
        CHECKERS[item].check()
