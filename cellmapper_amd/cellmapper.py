"""``CellMapper``: k-NN based transfer of labels, embeddings and expression -- MI355X-native drop-in.

Keeps the reference's user API (``cellmapper.py:20-637``): ``CellMapper(query, reference=None)``, ``.map(...)``,
``compute_neighbors`` / ``compute_mapping_matrix`` / ``map_obs`` / ``map_obsm`` / ``map_layers``,
``mapping_matrix`` (scipy CSR float32, settable), ``query_imputed``, ``load_precomputed_distances``.
All arithmetic (k-NN, graph kernel, normalisation, label vote, SpMM / SpGEMM) runs in the HIP backend.
The ``use_rep=None`` fallbacks (fast CCA / joint PCA) live in ``embedding.py``; the label evaluation mixin is out of scope.
"""
from __future__ import annotations

from typing import Any, Literal

import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix, issparse

from . import _native
from .embedding import EmbeddingMixin
from .evaluate import ExpressionEvaluationMixin, PresenceMixin
from .knn import Neighbors
from .logging import logger
from .utils import create_imputed_anndata, get_n_comps

_KERNEL_METHODS = ("gaussian", "scarches", "inverse_distance", "random", "equal")


class CellMapper(PresenceMixin, ExpressionEvaluationMixin, EmbeddingMixin):
    """Mapping of labels, embeddings, and expression values between reference and query datasets."""

    def __init__(self, query, reference=None) -> None:
        self.query = query
        self.reference = reference if reference is not None else query
        self._is_self_mapping = reference is None
        if self._is_self_mapping:
            logger.info("Initialized CellMapper for self-mapping with %d cells.", query.n_obs)
        else:
            logger.info("Initialized CellMapper with %d query cells and %d reference cells.", query.n_obs,
                        self.reference.n_obs)
        self.knn: Neighbors | None = None
        self._mapping_matrix: csr_matrix | None = None
        self._mapping_graph = None   # device-resident mapping matrix (a DeviceGraph); scipy CSR is built on first read
        self._mapping_tag = None     # (kernel, epsilon) that matrix was made with
        self.label_transfer_metrics: dict[str, Any] | None = None
        self.label_transfer_report: pd.DataFrame | None = None
        self.prediction_postfix: str | None = None
        self.confidence_postfix: str | None = None
        self.only_yx: bool | None = None
        self._query_imputed = None
        self.expression_transfer_metrics: dict[str, Any] | None = None

    def __repr__(self):
        q = f"AnnData(n_obs={self.query.n_obs:,}, n_vars={self.query.n_vars:,})"
        if self._is_self_mapping:
            return f"CellMapper(self-mapping, data={q})"
        r = f"AnnData(n_obs={self.reference.n_obs:,}, n_vars={self.reference.n_vars:,})"
        return f"CellMapper(query={q}, reference={r})"

    # ------------------------------------------------------------------ mapping matrix (K3) ----------
    @property
    def mapping_matrix(self):
        """Row-normalised mapping matrix, scipy CSR float32 of shape ``(n_query, n_reference)``.  After
        ``compute_mapping_matrix`` with a kernel method it lives on the device; the scipy object is built when read."""
        if self._mapping_graph is not None and self._mapping_graph.map_tag != self._mapping_tag:
            self._mapping_graph.mapping(*self._mapping_tag)   # (re-mapped in place by the shared NeighborsResults: restore)
            self._mapping_matrix = None
        if self._mapping_matrix is None and self._mapping_graph is not None:
            indptr, cols, vals = self._mapping_graph.fetch_mapping()
            m = csr_matrix((vals, cols, indptr), shape=(self.query.n_obs, self.reference.n_obs))
            m.has_sorted_indices = True
            m.has_canonical_format = True
            self._mapping_matrix = m
        return self._mapping_matrix

    @mapping_matrix.setter
    def mapping_matrix(self, value):
        self._mapping_graph = None   # a caller-supplied matrix replaces the resident one
        self._mapping_matrix = None if value is None else self._validate_and_normalize_mapping_matrix(value)

    def _has_mapping(self) -> bool:
        return self._mapping_matrix is not None or self._mapping_graph is not None

    def _validate_and_normalize_mapping_matrix(self, mapping_matrix) -> csr_matrix:
        """Shape check, float64 row normalisation on the GPU, CSR float32 (reference cellmapper.py:99-137)."""
        expected = (self.query.n_obs, self.reference.n_obs)
        if tuple(mapping_matrix.shape) != expected:
            raise ValueError(f"Mapping matrix shape mismatch: expected ({expected[0]}, {expected[1]}), "
                             f"but got {mapping_matrix.shape}.")
        m = csr_matrix(mapping_matrix, dtype=np.float64, copy=True)
        m.sum_duplicates()
        m.sort_indices()
        vals = _native.default_handle().csr_normalize(m.indptr, m.data)
        if m.nnz and bool(np.any(np.add.reduceat(np.abs(m.data), np.minimum(m.indptr[:-1], m.nnz - 1))[np.diff(m.indptr) > 0] == 0)):
            logger.warning("Some rows in the mapping matrix have a sum of zero. These rows will be left unchanged.")
        out = csr_matrix((vals, m.indices, m.indptr), shape=m.shape)
        return out

    # ------------------------------------------------------------------ K0 ---------------------------
    def compute_neighbors(
        self,
        n_neighbors: int = 30,
        use_rep: str | None = None,
        n_comps: int | None = None,
        method: Literal["hip", "sklearn", "pynndescent", "rapids", "faiss"] = "hip",
        metric: str = "euclidean",
        only_yx: bool = False,
        fallback_representation: Literal["fast_cca", "joint_pca"] = "fast_cca",
        fallback_kwargs: dict[str, Any] | None = None,
    ) -> None:
        """Nearest neighbors between reference and query in ``use_rep`` (reference cellmapper.py:139-251).

        ``use_rep="X"`` means ``.X`` (reserved, as upstream); otherwise ``.obsm[use_rep]``; the first
        ``min(n_comps or 50, d)`` columns are used.  ``use_rep=None`` computes a joint representation first
        (``fallback_representation``: fast CCA or joint PCA; plain PCA in self-mapping mode), see ``embedding.py``.
        """
        if fallback_kwargs is None:
            fallback_kwargs = {}
        else:
            fallback_kwargs = dict(fallback_kwargs)
        self.only_yx = only_yx
        if use_rep is None:   # joint representation computed on the fly (reference cellmapper.py:211-235)
            if self._is_self_mapping:
                logger.warning("No representation provided (`use_rep=None`) and self-mapping mode detected. Computing a "
                               "joint representation automatically using PCA.")
                key_added = fallback_kwargs.pop("key_added", "X_pca")
                from .embedding import joint_pca_scores

                X = self.query.X
                self.query.obsm[key_added] = joint_pca_scores(X if issparse(X) else np.asarray(X),
                                                              get_n_comps(n_comps, n_vars=X.shape[1]))
            else:
                logger.warning("No representation provided (`use_rep=None`). Computing a joint representation automatically "
                               "using '%s'. For optimal results, consider pre-computing a representation and passing it to "
                               "`use_rep`.", fallback_representation)
                if fallback_representation == "fast_cca":
                    key_added = fallback_kwargs.pop("key_added", "X_cca")
                    self.compute_fast_cca(n_comps=n_comps, key_added=key_added, **fallback_kwargs)
                elif fallback_representation == "joint_pca":
                    key_added = fallback_kwargs.pop("key_added", "X_pca")
                    self.compute_joint_pca(n_comps=n_comps, key_added=key_added, **fallback_kwargs)
                else:
                    raise ValueError(f"Unknown fallback_representation: {fallback_representation}. "
                                     "Supported options are 'fast_cca' and 'joint_pca'.")
            use_rep = key_added
        if use_rep == "X":
            xrep, yrep = self.reference.X, self.query.X
        else:
            xrep, yrep = self.reference.obsm[use_rep], self.query.obsm[use_rep]
        if issparse(xrep) or issparse(yrep):
            xrep = xrep.toarray() if issparse(xrep) else xrep
            yrep = yrep.toarray() if issparse(yrep) else yrep
        xrep, yrep = np.asarray(xrep), np.asarray(yrep)
        n_comps = get_n_comps(n_comps, n_vars=xrep.shape[1])
        xrep, yrep = xrep[:, :n_comps], yrep[:, :n_comps]
        self.knn = Neighbors(np.ascontiguousarray(xrep), np.ascontiguousarray(yrep))
        self.knn.compute_neighbors(n_neighbors=n_neighbors, method=method, metric=metric, only_yx=only_yx)

    # ------------------------------------------------------------------ K1-K3 ------------------------
    def compute_mapping_matrix(
        self,
        method: Literal["jaccard", "gaussian", "scarches", "inverse_distance", "random", "hnoca", "equal"] = "gaussian",
    ) -> None:
        """Mapping matrix from the query->reference graph (reference cellmapper.py:253-305)."""
        if self.knn is None:
            raise ValueError("Neighbors have not been computed. Call compute_neighbors() first.")
        logger.info("Computing mapping matrix using method '%s'.", method)
        if method in ("jaccard", "hnoca"):
            if self.only_yx:
                raise ValueError("Jaccard and HNOCa methods require both x and y neighbors to be computed. "
                                 "Set only_yx=False.")
            knn = self.knn
            if not (knn._has("xx") and knn._has("yy") and knn._has("xy") and knn._has("yx")):
                raise ValueError("Neighbors must be computed before accessing adjacency matrices.")
            # (yx @ xx.T) + (yy @ xy.T) on the 0/1 adjacencies + the jaccard / hnoca transform, on the GPU
            indptr, cols, vals = _native.default_handle().shared_neighbors(
                knn.yx.indices, knn.xx.indices, knn.yy.indices, knn.xy.indices, transform=method)
            self.mapping_matrix = csr_matrix((vals, cols, indptr), shape=(self.query.n_obs, self.reference.n_obs))
        elif method in _KERNEL_METHODS:
            if method == "random":
                self.mapping_matrix = self.knn.yx.knn_graph_connectivities(kernel="random")
                return
            yx = self.knn.yx
            expected = (self.query.n_obs, self.reference.n_obs)
            if yx.shape != expected:
                raise ValueError(f"Mapping matrix shape mismatch: expected ({expected[0]}, {expected[1]}), "
                                 f"but got {yx.shape}.")
            if getattr(yx, "_graph_dev", None) is not None:   # resident graph: K1-K3 on the device, nothing moves
                yx._graph_dev.mapping(method)
                self._mapping_graph, self._mapping_matrix = yx._graph_dev, None
                self._mapping_tag = yx._graph_dev.map_tag
            else:
                self._mapping_graph = None
                self._mapping_matrix = yx.normalized_mapping_matrix(kernel=method)
        else:
            raise NotImplementedError(f"Method '{method}' is not implemented.")

    # ------------------------------------------------------------------ A1 / A2 / A3 -----------------
    def _require_mapping(self):
        if not self._has_mapping():
            raise ValueError("Mapping matrix has not been computed. Call compute_mapping_matrix() first.")
        m = self.mapping_matrix
        return m.indptr.astype(np.int64, copy=False), m.indices, m.data

    def _backend(self):
        """(label_vote, spmm_dense, spmm_csr) bound to the resident mapping matrix when there is one, else to host arrays."""
        if not self._has_mapping():
            raise ValueError("Mapping matrix has not been computed. Call compute_mapping_matrix() first.")
        g = self._mapping_graph
        if g is not None:
            # the graph is shared with `self.knn.yx`: if somebody re-ran the mapping on it with another kernel
            # (NeighborsResults.normalized_mapping_matrix), put this mapper's own matrix back before using it
            if g.map_tag != self._mapping_tag:
                g.mapping(*self._mapping_tag)
                self._mapping_matrix = None
            return g.label_vote, g.spmm_dense, g.spmm_csr
        indptr, cols, vals = self._require_mapping()
        h = _native.default_handle()
        return (lambda codes, n_cat: h.label_vote(indptr, cols, vals, codes, n_cat),
                lambda B: h.spmm_dense(indptr, cols, vals, B),
                lambda bi, bc, bv, n_out: h.spmm_csr(indptr, cols, vals, bi, bc, bv, n_out))

    def map_obsm(self, key: str, prediction_postfix: str = "pred") -> None:
        """``query.obsm[f"{key}_{postfix}"] = M @ reference.obsm[key]`` (reference cellmapper.py:307-344)."""
        _, spmm_dense, _ = self._backend()
        logger.info("Mapping embeddings for key '%s'.", key)
        emb = np.asarray(self.reference.obsm[key])
        out_key = f"{key}_{prediction_postfix}"
        self.query.obsm[out_key] = spmm_dense(emb)
        logger.info("Embeddings mapped and stored in query.obsm['%s'].", out_key)

    def map_layers(self, key: str) -> None:
        """``query_imputed = M @ (reference.X | reference.layers[key])`` (reference cellmapper.py:346-383)."""
        _, spmm_dense, spmm_csr = self._backend()
        logger.info("Mapping layer for key '%s'.", key)
        layer = self.reference.X if key == "X" else self.reference.layers[key]
        if issparse(layer):
            B = csr_matrix(layer)
            if not B.has_canonical_format:
                B = B.copy()
                B.sum_duplicates()
            oi, oc, ov = spmm_csr(B.indptr, B.indices, B.data.astype(np.float32, copy=False), B.shape[1])
            res = csr_matrix((ov, oc, oi), shape=(len(oi) - 1, B.shape[1]))
        else:
            res = spmm_dense(np.asarray(layer))
        self.query_imputed = res
        msg = f"Expression for layer '{key}' mapped and stored in query_imputed.X."
        if not self._is_self_mapping:
            msg += (f"\nNote: The feature space matches the reference (n_vars={self.reference.n_vars}), "
                    f"not the query (n_vars={self.query.n_vars}).")
        logger.info(msg)

    @property
    def query_imputed(self):
        """Imputed query data as an AnnData-like object (or None)."""
        return self._query_imputed

    @query_imputed.setter
    def query_imputed(self, value) -> None:
        if value is None:
            self._query_imputed = None
            return
        self._query_imputed = create_imputed_anndata(expression_data=value, query_adata=self.query,
                                                     reference_adata=self.reference)

    # ------------------------------------------------------------------ orchestration ----------------
    def map(
        self,
        obs_keys: str | list[str] | None = None,
        obsm_keys: str | list[str] | None = None,
        layer_key: str | None = None,
        n_neighbors: int = 30,
        use_rep: str | None = None,
        knn_method: Literal["hip", "sklearn", "pynndescent", "rapids", "faiss"] = "hip",
        metric: str = "euclidean",
        only_yx: bool = False,
        mapping_method: Literal["jaccard", "gaussian", "scarches", "inverse_distance", "random", "hnoca", "equal"] = "gaussian",
        prediction_postfix: str = "pred",
    ) -> "CellMapper":
        """Neighbors -> mapping matrix -> transfer of obs / obsm / layer, in that order (reference :426-491)."""
        self.compute_neighbors(n_neighbors=n_neighbors, use_rep=use_rep, method=knn_method, metric=metric,
                               only_yx=only_yx)
        self.compute_mapping_matrix(method=mapping_method)
        if obs_keys is not None:
            for key in ([obs_keys] if isinstance(obs_keys, str) else obs_keys):
                self.map_obs(key=key, prediction_postfix=prediction_postfix)
        if obsm_keys is not None:
            for key in ([obsm_keys] if isinstance(obsm_keys, str) else obsm_keys):
                self.map_obsm(key=key, prediction_postfix=prediction_postfix)
        if layer_key is not None:
            self.map_layers(key=layer_key)
        if obs_keys is None and obsm_keys is None and layer_key is None:
            logger.warning("Neither ``obs_keys``, ``obsm_keys`` or ``layer_key`` provided. No labels, embeddings or "
                           "layers were transferred. Please provide at least one of ``obs_keys``, ``obsm_keys`` or "
                           "``layer_key``.")
        return self

    def load_precomputed_distances(self, distances_key: str = "distances", include_self: bool | None = None) -> None:
        """Use ``query.obsp[distances_key]`` instead of a search (self-mapping only; reference :493-532)."""
        if not self._is_self_mapping:
            raise ValueError("load_precomputed_distances is only available in self-mapping mode.")
        dm = self.query.obsp[distances_key]
        self.knn = Neighbors.from_distances(dm, include_self=include_self)
        logger.info("Loaded precomputed distances from '%s' with %d cells and %d neighbors per cell.", distances_key,
                    dm.shape[0], self.knn.xx.n_neighbors)

    # ------------------------------------------------------------------ A0 / A0n ---------------------
    def map_obs(self, key: str, prediction_postfix: str = "pred", confidence_postfix: str = "conf") -> None:
        """Transfer ``reference.obs[key]``: weighted vote for categorical data, weighted mean for numeric
        (reference cellmapper.py:534-637)."""
        if not self._has_mapping():
            raise ValueError("Mapping matrix has not been computed. Call compute_mapping_matrix() first.")
        if key not in self.reference.obs.columns:
            raise KeyError(f"Key '{key}' not found in reference.obs")
        self.prediction_postfix = prediction_postfix
        self.confidence_postfix = confidence_postfix
        col = self.reference.obs[key]
        categorical = (isinstance(col.dtype, pd.CategoricalDtype) or pd.api.types.is_object_dtype(col)
                       or pd.api.types.is_string_dtype(col))
        if categorical:
            logger.info("Mapping categorical data for key '%s' using one-hot encoding.", key)
            self._map_obs_categorical(key, prediction_postfix, confidence_postfix)
        else:
            logger.info("Mapping numerical data for key '%s' using direct matrix multiplication.", key)
            self._map_obs_numerical(key, prediction_postfix)

    def _map_obs_categorical(self, key: str, prediction_postfix: str, confidence_postfix: str) -> None:
        label_vote, _, _ = self._backend()
        col = self.reference.obs[key]
        # sorted unique OBSERVED values: the category order sklearn's OneHotEncoder would produce
        if isinstance(col.dtype, pd.CategoricalDtype) and not col.isna().any():
            raw = col.cat.codes.to_numpy()
            cats = np.asarray(col.cat.categories)
            present = np.flatnonzero(np.bincount(raw, minlength=len(cats)))   # observed categories (no NaN here: codes >= 0)
            order = present[np.argsort(cats[present], kind="stable")]
            remap = np.full(len(cats), -1, dtype=np.int32)
            remap[order] = np.arange(len(order))
            codes, categories = remap[raw], cats[order]
            cat_order = order
        else:
            categories, codes = np.unique(np.asarray(col), return_inverse=True)
            cat_order = None
        code, conf = label_vote(codes.astype(np.int32), len(categories))
        if cat_order is not None:   # categorical column: straight from codes (no factorisation of n_query strings)
            pred = pd.Series(pd.Categorical.from_codes(cat_order[code], dtype=col.dtype), index=self.query.obs_names)
        else:
            pred = pd.Series(data=categories[code], index=self.query.obs_names, dtype=col.dtype)
        self.query.obs[f"{key}_{prediction_postfix}"] = pred
        self.query.obs[f"{key}_{confidence_postfix}"] = pd.Series(conf, index=self.query.obs_names)
        if f"{key}_colors" in self.reference.uns and isinstance(col.dtype, pd.CategoricalDtype):
            lookup = dict(zip(col.cat.categories, self.reference.uns[f"{key}_colors"], strict=True))
            self.query.uns[f"{key}_{prediction_postfix}_colors"] = [lookup.get(c, "#383838") for c in pred.cat.categories]
        logger.info("Categorical data mapped and stored in query.obs['%s'].", f"{key}_{prediction_postfix}")
        del codes, code, conf

    def _map_obs_numerical(self, key: str, prediction_postfix: str) -> None:
        _, spmm_dense, _ = self._backend()
        values = np.array(self.reference.obs[key]).reshape(-1, 1)
        mapped = spmm_dense(values)
        self.query.obs[f"{key}_{prediction_postfix}"] = pd.Series(data=mapped.ravel(), index=self.query.obs_names)
        logger.info("Numerical data mapped and stored in query.obs['%s'].", f"{key}_{prediction_postfix}")
