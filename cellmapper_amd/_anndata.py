"""Minimal AnnData-compatible host container.

The reference operates on ``anndata.AnnData`` objects (``cellmapper.py:23-59``) but only
touches a small duck-typed surface of them on the hot path: ``.X``, ``.obs`` (DataFrame),
``.var``, ``.obsm``/``.varm``/``.layers``/``.obsp`` (mappings), ``.uns``, ``.n_obs``,
``.n_vars``, ``.obs_names``, ``.var_names``.  When the real ``anndata`` package is
importable we re-export its class; otherwise this small container stands in so that the
drop-in ``CellMapper`` works (and is testable) in environments without anndata.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

try:  # pragma: no cover - exercised only where anndata is installed
    from anndata import AnnData as _RealAnnData  # type: ignore
except Exception:  # ModuleNotFoundError or a broken install
    _RealAnnData = None


class MiniAnnData:
    """Duck-typed stand-in for ``anndata.AnnData`` (host-side container glue only)."""

    def __init__(self, X=None, obs=None, var=None, uns=None, obsm=None, varm=None, layers=None, obsp=None):
        n_obs = n_vars = None
        if X is not None:
            n_obs, n_vars = X.shape
        if obs is None:
            if n_obs is None:
                raise ValueError("Either X or obs must be given.")
            obs = pd.DataFrame(index=pd.RangeIndex(n_obs).astype(str))
        if var is None:
            var = pd.DataFrame(index=pd.RangeIndex(n_vars if n_vars is not None else 0).astype(str))
        if X is not None and (len(obs) != n_obs or len(var) != n_vars):
            raise ValueError(
                f"X has shape {X.shape} but obs has {len(obs)} rows and var has {len(var)} rows."
            )
        self.X = X
        self.obs = obs
        self.var = var
        self.uns = uns if uns is not None else {}
        self.obsm = obsm if obsm is not None else {}
        self.varm = varm if varm is not None else {}
        self.layers = layers if layers is not None else {}
        self.obsp = obsp if obsp is not None else {}

    @property
    def n_obs(self) -> int:
        return len(self.obs)

    @property
    def n_vars(self) -> int:
        return len(self.var)

    @property
    def shape(self):
        return (self.n_obs, self.n_vars)

    @property
    def obs_names(self) -> pd.Index:
        return self.obs.index

    @property
    def var_names(self) -> pd.Index:
        return self.var.index

    def copy(self) -> "MiniAnnData":
        cp = lambda v: v.copy() if hasattr(v, "copy") else v  # noqa: E731
        return MiniAnnData(
            X=cp(self.X) if self.X is not None else None,
            obs=self.obs.copy(),
            var=self.var.copy(),
            uns=dict(self.uns),
            obsm={k: cp(v) for k, v in self.obsm.items()},
            varm={k: cp(v) for k, v in self.varm.items()},
            layers={k: cp(v) for k, v in self.layers.items()},
            obsp={k: cp(v) for k, v in self.obsp.items()},
        )

    def __repr__(self) -> str:
        return f"MiniAnnData(n_obs={self.n_obs}, n_vars={self.n_vars}, obsm={list(self.obsm)}, layers={list(self.layers)})"


AnnData = _RealAnnData if _RealAnnData is not None else MiniAnnData


def is_anndata_like(obj) -> bool:
    """True for real AnnData, MiniAnnData, or any object exposing the duck-typed surface."""
    return all(hasattr(obj, a) for a in ("obs", "var", "obsm", "n_obs", "n_vars", "X"))


def make_anndata(X, obs, var, uns, obsm, varm=None):
    """Construct an AnnData-like object from parts (real class when available)."""
    if _RealAnnData is not None and isinstance(obs, pd.DataFrame):
        try:
            return _RealAnnData(X=X, obs=obs, var=var, uns=uns, obsm=obsm, varm=varm)
        except Exception:  # fall back to the shim for exotic inputs
            pass
    return MiniAnnData(X=X, obs=obs, var=var, uns=uns, obsm=obsm, varm=varm)


__all__ = ["AnnData", "MiniAnnData", "is_anndata_like", "make_anndata", "np"]
