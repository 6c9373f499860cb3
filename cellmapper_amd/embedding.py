"""Joint representations for ``use_rep=None``: fast CCA and joint PCA on the MI355X backend.

Host-side mirror of the reference's ``EmbeddingMixin`` (``model/embedding.py:11-230``) and of
``truncated_svd_cross_covariance`` (``utils.py:230-398``): same method names, arguments, defaults and side effects
(``.obsm[key_added]`` of both objects, ``.uns[f"{key_added}_params"]``).  The reference runs ARPACK on an implicit
operator ``v -> (X - rowmean)((Y - rowmean)^T v) / (G - 1)``; here the same singular triplets come from two float64 Gram
matrices computed on the f64 matrix cores (``cm_gram``), their p x p eigen-decompositions and one small SVD on the host
(LAPACK), and two projections back onto the cells (``cm_project``):

    Xc = Qx diag(lx) Ax^T,  Yc = Qy diag(ly) Ay^T        (Ax, lx^2: eigenpairs of Xc^T Xc; likewise Y)
    Xc Yc^T / (G-1) = Qx [ diag(lx) Ax^T Ay diag(ly) / (G-1) ] Qy^T = Qx (Um S Vm^T) Qy^T
    U = Xc Ax diag(1/lx) Um,   V = Yc Ay diag(1/ly) Vm

Exact (no iteration, no random start), so results are deterministic; they agree with the reference's up to the sign of
each singular vector and ARPACK's tolerance.  Everything of size cells x genes is touched only by the two kernels.
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix, issparse

from . import _native
from .logging import logger
from .utils import get_n_comps

_RANK_TOL = 1e-11   # eigenvalues of a Gram matrix below this fraction of the largest are treated as zero


def _factor(gram: np.ndarray):
    """(A, l): Xc = Q diag(l) A^T with orthonormal Q, from the eigen-decomposition of Xc^T Xc."""
    w, E = np.linalg.eigh(gram)
    keep = w > _RANK_TOL * max(w[-1], 0.0)
    return E[:, keep], np.sqrt(w[keep])


def truncated_svd_cross_covariance(X, Y, n_comps: int = 50, zero_center: bool = True, implicit: bool = True,
                                   random_state: int = 0):
    """Truncated SVD of the cross-covariance ``X' Y'^T / (G - 1)`` (reference utils.py:230-398; ``X'`` = rows centred at
    their means when ``zero_center``).  Returns ``(U [n_x, n_comps], s [n_comps], Vt [n_comps, n_y])``, singular values
    descending.  ``implicit`` / ``random_state`` are accepted for signature compatibility (the computation is exact)."""
    if X.shape[1] != Y.shape[1]:
        raise ValueError(f"X and Y must have the same number of variables: X has {X.shape[1]}, Y has {Y.shape[1]}")
    if issparse(X) != issparse(Y):
        raise TypeError("X and Y must be of the same type: both sparse or both dense")
    h = _native.default_handle()
    G = X.shape[1]
    scale = 1.0 / (G - 1)
    Ax, lx = _factor(h.gram(X, center_rows=zero_center))
    Ay, ly = _factor(h.gram(Y, center_rows=zero_center))
    M = scale * (lx[:, None] * (Ax.T @ Ay)) * ly[None, :]
    Um, s, Vmt = np.linalg.svd(M, full_matrices=False)
    nc = min(n_comps, len(s))
    if nc < n_comps:
        logger.warning("Cross-covariance has rank %d < n_comps=%d; returning %d components.", len(s), n_comps, nc)
    Px = Ax @ (Um[:, :nc] / lx[:, None])
    Py = Ay @ (Vmt[:nc].T / ly[:, None])
    U = h.project(X, Px, center_rows=zero_center)
    V = h.project(Y, Py, center_rows=zero_center)
    return U, s[:nc], V.T


def joint_pca_scores(A, n_comps: int):
    """Scores of the top ``n_comps`` principal components of the cells x genes matrix ``A`` (columns centred), sklearn's
    ``PCA(svd_solver="arpack").fit_transform`` -- what ``scanpy.pp.pca`` runs -- with sklearn's sign convention (the entry
    of largest magnitude of every component is positive)."""
    h = _native.default_handle()
    n = A.shape[0]
    mu = np.asarray(A.sum(axis=0, dtype=np.float64)).ravel() / n   # (float64 accumulation: float32 sums cost 1e-5 in the scores)
    g = h.gram(A, center_rows=False) - n * np.outer(mu, mu)
    w, E = np.linalg.eigh(g)
    W = E[:, ::-1][:, :n_comps]
    idx = np.argmax(np.abs(W), axis=0)
    W = W * np.sign(W[idx, np.arange(W.shape[1])])[None, :]
    return h.project(A, W, col_shift=mu)


def _inner_join(reference, query):
    """Positions of the shared variables in both objects, in the reference's order (``ad.concat(join="inner")``)."""
    common = reference.var_names.intersection(query.var_names, sort=False)
    return common, reference.var_names.get_indexer(common), query.var_names.get_indexer(common)


def _take_columns(X, idx):
    if issparse(X):
        return csr_matrix(X)[:, idx]
    return np.asarray(X)[:, idx]


class EmbeddingMixin:
    """Joint embeddings for two datasets (reference ``model/embedding.py``)."""

    def compute_joint_pca(self, n_comps: int | None = None, key_added: str = "X_pca", **kwargs) -> None:
        """Joint PCA on the ``.X`` matrices of reference and query over their shared genes (reference :14-48)."""
        if kwargs:
            logger.warning("compute_joint_pca: scanpy-specific options %s are ignored by the hip backend.", sorted(kwargs))
        common, ir, iq = _inner_join(self.reference, self.query)
        Xr, Xq = _take_columns(self.reference.X, ir), _take_columns(self.query.X, iq)
        if issparse(Xr) != issparse(Xq):
            Xr = Xr.toarray() if issparse(Xr) else Xr
            Xq = Xq.toarray() if issparse(Xq) else Xq
        if issparse(Xr):
            from scipy.sparse import vstack

            joint = vstack([Xr, Xq]).tocsr()
        else:
            joint = np.concatenate([Xr, Xq], axis=0)
        n_comps = get_n_comps(n_comps, n_vars=len(common))
        Z = joint_pca_scores(joint, n_comps)
        n_ref = self.reference.n_obs
        self._set_embedding(X_query=Z[n_ref:], X_ref=Z[:n_ref], key_added=key_added, method="fast_cca", n_comps=n_comps,
                            n_common_genes=len(common))   # (method label as upstream, embedding.py:46)

    def compute_fast_cca(self, n_comps: int | None = None, key_added: str = "X_cca", layer: str | None = None,
                         mask_var=None, zero_center: bool = True, scale_with_singular: bool = False, l2_scale: bool = True,
                         random_state: int = 0, implicit: bool = True) -> None:
        """Joint embedding by fast CCA (reference :50-185): SVD of the cross-covariance between query and reference over
        their shared genes, singular vectors optionally scaled by sqrt(s) and l2-normalised per cell."""
        logger.info("Computing fast CCA between query (%d cells) and reference (%d cells).", self.query.n_obs,
                    self.reference.n_obs)
        common, ir, iq = _inner_join(self.reference, self.query)
        if mask_var is not None:
            if isinstance(mask_var, str):
                m = np.asarray(self.reference.var[mask_var].to_numpy()[ir], bool) & np.asarray(self.query.var[mask_var].to_numpy()[iq], bool)
            else:
                m = np.asarray(mask_var, bool)
                if m.shape[0] != len(common):
                    raise ValueError(f"mask_var has {m.shape[0]} entries for {len(common)} shared variables.")
            common, ir, iq = common[m], ir[m], iq[m]
        src_q = self.query.X if layer is None else self.query.layers[layer]
        src_r = self.reference.X if layer is None else self.reference.layers[layer]
        X_query, X_ref = _take_columns(src_q, iq), _take_columns(src_r, ir)
        n_common_genes = len(common)
        logger.info("Using %d common genes between query and reference datasets.", n_common_genes)
        n_comps = get_n_comps(n_comps, n_vars=n_common_genes)
        if issparse(X_query) != issparse(X_ref):
            logger.info("Converting matrices to ensure consistent type (both sparse or both dense).")
            if issparse(X_query):
                X_ref = csr_matrix(X_ref)
            else:
                X_query = X_query.toarray() if hasattr(X_query, "toarray") else X_query
        U, s, Vt = truncated_svd_cross_covariance(X_query, X_ref, n_comps=n_comps, zero_center=zero_center,
                                                  random_state=random_state, implicit=implicit)
        V = Vt.T
        logger.info("SVD of cross-covariance matrix computed successfully.")
        if scale_with_singular:
            U = U * np.sqrt(s)[None, :]
            V = V * np.sqrt(s)[None, :]
        if l2_scale:
            U = U / np.linalg.norm(U, axis=1)[:, None]
            V = V / np.linalg.norm(V, axis=1)[:, None]
        self._set_embedding(X_query=U, X_ref=V, key_added=key_added, method="fast_cca", n_comps=n_comps,
                            n_common_genes=n_common_genes)

    def _set_embedding(self, X_query, X_ref, key_added: str, method: str, n_comps: int, n_common_genes: int) -> None:
        """Store the joint embedding and its parameters in both objects (reference :187-230)."""
        self.query.obsm[key_added] = X_query
        self.reference.obsm[key_added] = X_ref
        logger.info("Joint embedding stored as '%s' in both reference.obsm and query.obsm.", key_added)
        params = {"n_comps": n_comps, "n_common_genes": n_common_genes, "method": method}
        self.reference.uns[f"{key_added}_params"] = params
        self.query.uns[f"{key_added}_params"] = params
        logger.info("Embedding computed with method '%s' stored as '%s' in both reference.obsm and query.obsm. ", method,
                    key_added)
