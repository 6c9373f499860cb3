"""Neighbor graph + graph kernel: the host-side mirror of the reference's ``model/knn.py``.

Same names, arguments, defaults, side effects and error behaviour as the reference
(``NeighborsResults`` knn.py:14-266, ``Neighbors`` knn.py:269-492), but every number is produced by the HIP
backend through the C ABI (``include/cellmap.h``); nothing on the hot path is computed with numpy / scipy /
sklearn here, and there is no CPU fallback.
"""
from __future__ import annotations

from functools import cached_property
from typing import Literal

import numpy as np
from scipy.sparse import csr_matrix

from . import _native
from .check import KNN_METHODS, check_deps
from .logging import logger
from .utils import extract_neighbors_from_distances

_KERNELS = ("gaussian", "scarches", "random", "inverse_distance", "equal")


def _csr(indptr, cols, data, shape, dtype):
    m = csr_matrix((np.asarray(data).astype(dtype, copy=False), cols, indptr), shape=shape)
    m.has_sorted_indices = True
    m.has_canonical_format = True
    return m


class NeighborsResults:
    """Nearest-neighbor results store (reference knn.py:14-66).

    ``distances`` / ``indices`` have shape ``(n_samples, n_neighbors)``; padded entries are ``inf`` / ``-1``.
    ``n_targets`` is the number of columns of the graph matrices (defaults to ``n_samples``).

    Results of :meth:`Neighbors.compute_neighbors` are DEVICE-RESIDENT (``_graph``: a ``cm_graph`` of the C ABI): the two
    arrays are copied to the host the first time ``distances`` / ``indices`` is read, so the staged flow
    ``compute_neighbors -> compute_mapping_matrix -> map_*`` never moves them at all.
    """

    def __init__(self, distances: np.ndarray | None = None, indices: np.ndarray | None = None, n_targets: int | None = None,
                 *, _graph=None):
        self._graph_dev = _graph
        self._distances, self._indices = distances, indices
        if _graph is None:
            if distances is None or indices is None:
                raise TypeError("NeighborsResults needs `distances` and `indices`.")
            if self._indices.shape != self._distances.shape:
                raise ValueError("Indices and distances must have the same shape.")
            self._shape2 = tuple(self._indices.shape)
        else:
            self._shape2 = (_graph.nq, _graph.k)
        self.n_targets = self._shape2[0] if n_targets is None else n_targets

    def _materialize(self):
        if self._distances is None or self._indices is None:
            self._distances, self._indices = self._graph_dev.fetch_knn()

    @property
    def distances(self) -> np.ndarray:
        self._materialize()
        return self._distances

    @property
    def indices(self) -> np.ndarray:
        self._materialize()
        return self._indices

    def __repr__(self):
        where = "device" if (self._graph_dev is not None and self._distances is None) else "host"
        return f"NeighborsResults(n_samples={self._shape2[0]}, n_neighbors={self._shape2[1]}, n_targets={self.n_targets}, data={where})"

    def __eq__(self, other):
        return (isinstance(other, NeighborsResults) and self.n_targets == other.n_targets
                and np.array_equal(self.distances, other.distances) and np.array_equal(self.indices, other.indices))

    @property
    def n_samples(self) -> int:
        return self._shape2[0]

    @property
    def n_neighbors(self) -> int:
        return self._shape2[1]

    @property
    def shape(self) -> tuple[int, int]:
        return (self.n_samples, self.n_targets or self.n_samples)

    def _get_valid_entries_mask(self) -> np.ndarray:
        """``(indices != -1) & isfinite(distances)`` (reference knn.py:68-77); informational only."""
        return (self.indices != -1) & np.isfinite(self.distances)

    # -- graph matrices: K1 + K2 on the GPU (cm_mapping_matrix) ---------------------------------------
    def _graph(self, kernel: str, dist=None, epsilon: float = 1e-8):
        h = _native.default_handle()
        d = self.distances if dist is None else dist
        return h.mapping_matrix(d, self.indices, self.shape[1], kernel, epsilon, want_conn=True)

    @cached_property
    def knn_graph_distances(self) -> csr_matrix:
        """Sparse matrix of neighbor distances, shape ``(n_samples, n_targets)`` (reference knn.py:113-132)."""
        indptr, cols, _, conn, _ = self._graph("_identity")
        return _csr(indptr, cols, conn, self.shape, np.float64)

    def knn_graph_connectivities(
        self,
        kernel: Literal["gaussian", "scarches", "random", "inverse_distance", "equal"] = "gaussian",
        dtype=np.float64,
        **kwargs,
    ) -> csr_matrix:
        """Connectivities with the given kernel as canonical CSR (reference knn.py:134-226)."""
        if kernel not in _KERNELS:
            raise ValueError(
                f"Unknown kernel: {kernel}. Supported kernels are: 'gaussian', 'scarches', 'random', "
                "'inverse_distance', 'equal'.")
        if kernel == "random":  # testing aid upstream (np.random.rand on the valid entries)
            valid = self._get_valid_entries_mask()
            if not valid.any():
                raise ValueError("No finite distances found in the neighborhood graph")
            w = np.where(valid, np.random.rand(*self.distances.shape), np.inf)
            indptr, cols, _, conn, _ = self._graph("_identity", dist=w)
        else:
            indptr, cols, _, conn, _ = self._graph(kernel, epsilon=kwargs.get("epsilon", 1e-8))
        return _csr(indptr, cols, conn, self.shape, dtype)

    def boolean_adjacency(self, dtype=np.float64, set_diag: bool | None = None) -> csr_matrix:
        """0/1 adjacency from the indices alone (reference knn.py:228-266)."""
        zeros = np.zeros(self.indices.shape, np.float64)  # the mask here ignores distances
        indptr, cols, _, conn, _ = self._graph("equal", dist=zeros)
        adj = _csr(indptr, cols, conn, self.shape, dtype)
        if set_diag is not None:
            if self.shape[0] != self.shape[1]:
                raise ValueError("The set_diag parameter can only be used with square matrices "
                                 f"(got shape {self.shape[0]} x {self.shape[1]}).")
            adj = adj.tolil()
            adj.setdiag(1.0 if set_diag else 0.0)
            adj = adj.tocsr()
            if not set_diag:
                adj.eliminate_zeros()
        return adj

    def normalized_mapping_matrix(self, kernel: str = "gaussian", **kwargs) -> csr_matrix:
        """Kernel weights -> canonical CSR -> row-normalised float32 in ONE device pass (K1+K2+K3)."""
        if kernel == "random":
            return None
        if self._graph_dev is not None:   # resident graph: nothing is uploaded
            self._graph_dev.mapping(kernel, kwargs.get("epsilon", 1e-8))
            indptr, cols, vals = self._graph_dev.fetch_mapping()
            return _csr(indptr, cols, vals, self.shape, np.float32)
        h = _native.default_handle()
        indptr, cols, vals, _, _ = h.mapping_matrix(self.distances, self.indices, self.shape[1], kernel,
                                                   kwargs.get("epsilon", 1e-8))
        return _csr(indptr, cols, vals, self.shape, np.float32)


class Neighbors:
    """Compute and store the k-NN graphs between a reference (``xrep``) and a query (``yrep``) representation."""

    def __init__(self, xrep: np.ndarray, yrep: np.ndarray | None = None):
        self.xrep = xrep
        self.yrep = yrep if yrep is not None else xrep
        self._res: dict = {"xx": None, "yy": None, "xy": None, "yx": None}
        self._pending: dict | None = None   # search parameters of the graphs that are computed on first access (see compute_neighbors)
        self._is_self_mapping = yrep is None

    # The four graphs.  Upstream `compute_neighbors(only_yx=False)` -- the DEFAULT of `CellMapper.map()` -- runs four searches
    # (knn.py:441-457), of which the kernel mapping methods, every `map_*` transfer and the presence score read `yx` alone; only
    # jaccard / hnoca and `get_adjacency_matrices` read the other three.  Here `yx` is searched at once and `xx` / `yy` / `xy` on
    # their first access (same arrays, same errors -- the size check that would have failed upstream is made up front): the
    # default `map()` of a 1.5 M x 1.5 M problem does one search instead of four.
    def _get(self, name: str):
        if self._res[name] is None and self._pending is not None and name != "yx":
            self._compute_graph(name)
        return self._res[name]

    def _has(self, name: str) -> bool:
        return self._res[name] is not None or (self._pending is not None and name != "yx")

    xx = property(lambda self: self._get("xx"), lambda self, v: self._res.__setitem__("xx", v))
    yy = property(lambda self: self._get("yy"), lambda self, v: self._res.__setitem__("yy", v))
    xy = property(lambda self: self._get("xy"), lambda self, v: self._res.__setitem__("xy", v))
    yx = property(lambda self: self._get("yx"), lambda self, v: self._res.__setitem__("yx", v))

    def _compute_graph(self, name: str):
        p = self._pending
        h = _native.default_handle()
        ref_host, qry_host = {"xx": (self.xrep, self.xrep), "yy": (self.yrep, self.yrep), "xy": (self.yrep, self.xrep)}[name]
        ref = _native.DeviceRep(h, np.asarray(ref_host).astype(p["dt"], copy=False), allow_nan=p["nan_ok"])
        qry = ref if qry_host is ref_host else _native.DeviceRep(h, np.asarray(qry_host).astype(p["dt"], copy=False), allow_nan=p["nan_ok"])
        try:
            n_targets = None if name in ("xx", "yy") else ref_host.shape[0]
            self._res[name] = NeighborsResults(n_targets=n_targets, _graph=_native.DeviceGraph(h, ref, qry, p["k"], p["metric"]))
        finally:
            ref.close()
            if qry is not ref:
                qry.close()

    @classmethod
    def from_distances(cls, distances_matrix, include_self: bool | None = None) -> "Neighbors":
        """Build from a precomputed square sparse distance matrix (reference knn.py:296-337)."""
        indices, distances = extract_neighbors_from_distances(distances_matrix, include_self=include_self)
        n_cells = distances_matrix.shape[0]
        nb = cls(xrep=np.zeros((n_cells, 1)))
        res = NeighborsResults(distances=distances, indices=indices)
        nb.xx = nb.yy = nb.xy = nb.yx = res
        nb._is_self_mapping = True
        logger.info("Created Neighbors object from distances matrix with %d cells", n_cells)
        return nb

    def compute_neighbors(
        self,
        n_neighbors: int = 30,
        method: Literal["hip", "sklearn", "pynndescent", "rapids", "faiss"] = "hip",
        metric: str = "euclidean",
        random_state: int = 0,
        only_yx: bool = False,
    ):
        """Exact k-NN on the MI355X backend (reference knn.py:339-465).

        ``method`` accepts the reference's backend names for drop-in compatibility; all of them are served by
        the exact HIP search, whose results follow sklearn's brute-force semantics (true Euclidean distances,
        float64; cosine in the input dtype).  ``random_state`` is unused (the search is exact).
        """
        if method not in KNN_METHODS:
            raise ValueError(f"Unknown method: {method}. Supported methods are 'sklearn', 'pynndescent', and 'rapids'.")
        check_deps("hip")
        logger.info("Using %s to compute %d neighbors.", "hip" if method == "hip" else f"hip (requested: {method})",
                    n_neighbors)
        h = _native.default_handle()
        if metric not in _native.METRICS:
            raise ValueError(f"Metric '{metric}' is not supported by the hip backend (supported: {_native.SUPPORTED_METRICS_TEXT}).")
        if 9 <= _native.METRICS[metric] <= 15 and (self.xrep.dtype != bool or self.yrep.dtype != bool):
            # the boolean metrics read `x != 0`; sklearn says so (metrics/pairwise.py: DataConversionWarning, a UserWarning)
            import warnings
            warnings.warn(f"Data was converted to boolean for metric {metric}", UserWarning, stacklevel=2)
        nx, ny = self.xrep.shape[0], self.yrep.shape[0]
        # both representations go to the device ONCE (same dtype rule as the host tier: float32 only if both are);
        # the results stay there (NeighborsResults materialises numpy arrays on first access)
        dt = np.float32 if (self.xrep.dtype == np.float32 and self.yrep.dtype == np.float32) else np.float64
        nan_ok = metric == "nan_euclidean"   # (sklearn validates with ensure_all_finite="allow-nan" under this metric only)
        xr = _native.DeviceRep(h, np.asarray(self.xrep).astype(dt, copy=False), allow_nan=nan_ok)
        yr = xr if self.yrep is self.xrep else _native.DeviceRep(h, np.asarray(self.yrep).astype(dt, copy=False), allow_nan=nan_ok)
        self._pending = None
        self._res = {"xx": None, "yy": None, "xy": None, "yx": None}
        if not only_yx and n_neighbors > min(nx, ny):
            # (the yy / xy searches upstream would raise this before anything is returned: sklearn neighbors/_base.py:848-859)
            raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {n_neighbors}, n_samples_fit = {min(nx, ny)}")
        try:
            self.yx = NeighborsResults(n_targets=nx, _graph=_native.DeviceGraph(h, xr, yr, n_neighbors, metric))
        finally:
            xr.close()
            if yr is not xr:
                yr.close()
        if not only_yx:   # xx / yy / xy: on first access (see the properties above)
            self._pending = dict(k=int(n_neighbors), metric=metric, dt=dt, nan_ok=nan_ok)

    def get_adjacency_matrices(self):
        """Unweighted adjacency matrices ``(xx, yy, xy, yx)`` (reference knn.py:467-483)."""
        if not (self._has("xx") and self._has("yy") and self._has("xy") and self._has("yx")):
            raise ValueError("Neighbors must be computed before accessing adjacency matrices.")
        return (self.xx.boolean_adjacency(), self.yy.boolean_adjacency(), self.xy.boolean_adjacency(),
                self.yx.boolean_adjacency())

    def __repr__(self):
        return (f"Neighbors(xrep_shape={self.xrep.shape}, yrep_shape={self.yrep.shape}, "
                f"xx={self._has('xx')}, yy={self._has('yy')}, "
                f"xy={self._has('xy')}, yx={self._has('yx')}, "
                f"self_mapping={self._is_self_mapping})")
