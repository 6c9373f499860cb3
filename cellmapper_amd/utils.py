"""Host-side glue used by the boundary (container handling only; no arithmetic of the hot path lives here)."""
from __future__ import annotations

import numpy as np
import pandas as pd
from scipy.sparse import csr_matrix, issparse

from ._anndata import is_anndata_like, make_anndata
from .constants import PackageConstants
from .logging import logger


def get_n_comps(n_comps: int | None, n_vars: int) -> int:
    """Number of representation columns to use: ``min(n_comps or 50, n_vars)`` (reference utils.py:223-227)."""
    return min(n_vars, PackageConstants.n_comps if n_comps is None else n_comps)


def create_imputed_anndata(expression_data, query_adata, reference_adata):
    """Wrap transferred expression as an AnnData-like object: observations from the query, features from the
    reference (reference utils.py:15-126 -- same accepted types, shape checks and error classes)."""
    if is_anndata_like(expression_data) and not isinstance(expression_data, (np.ndarray, pd.DataFrame)):
        if expression_data.n_obs != query_adata.n_obs:
            raise ValueError(
                f"Imputed AnnData has {expression_data.n_obs} observations, but query has {query_adata.n_obs} "
                "observations. They must have the same number of observations.")
        if not expression_data.obs_names.equals(query_adata.obs_names):
            logger.warning("Observation names in imputed AnnData don't match query observation names. "
                           "Make sure the cells are aligned correctly.")
        return expression_data
    if not (isinstance(expression_data, (np.ndarray, pd.DataFrame)) or issparse(expression_data)):
        raise TypeError(f"Unsupported type for expression_data: {type(expression_data)}. "
                        "Must be AnnData, numpy array, sparse matrix, or pandas DataFrame.")
    if isinstance(expression_data, pd.DataFrame):
        if len(expression_data.index) != query_adata.n_obs:
            raise ValueError(f"DataFrame has {len(expression_data.index)} rows, but query has {query_adata.n_obs} "
                             "observations. They must match.")
        if len(expression_data.columns) != reference_adata.n_vars:
            raise ValueError(f"DataFrame has {len(expression_data.columns)} columns, but reference has "
                             f"{reference_adata.n_vars} features. They must match.")
        expression_data = expression_data.values
    expected = (query_adata.n_obs, reference_adata.n_vars)
    if tuple(expression_data.shape) != expected:
        raise ValueError(f"Expression data shape mismatch: expected {expected}, but got {expression_data.shape}. "
                         "Should be (n_query_cells, n_reference_genes).")
    varm = getattr(reference_adata, "varm", None)
    out = make_anndata(X=expression_data, obs=query_adata.obs, var=reference_adata.var, uns=dict(query_adata.uns),
                       obsm=query_adata.obsm, varm=varm)
    logger.info("Imputed expression matrix with shape %s converted to AnnData object.", expression_data.shape)
    return out


def extract_neighbors_from_distances(distances_matrix, include_self: bool | None = None):
    """Square sparse distance matrix (``adata.obsp['distances']``) -> padded ``(indices, distances)`` arrays
    with ``-1`` / ``inf`` padding, rows sorted by distance (contract of reference utils.py:129-219).

    Runs on the GPU (``cm_ingest_distances``: one wave per row, in-LDS stable rank sort for rows that are not already
    sorted) instead of two Python loops over all cells; the argument checks mirror the reference's.
    """
    if not issparse(distances_matrix):
        raise TypeError("Distances matrix must be a sparse matrix")
    if distances_matrix.shape[0] != distances_matrix.shape[1]:
        raise ValueError(f"Square distance matrix required (got {distances_matrix.shape})")
    m = csr_matrix(distances_matrix)
    from . import _native

    return _native.default_handle().ingest_distances(m.indptr, m.indices, m.data, m.shape[0], include_self)
