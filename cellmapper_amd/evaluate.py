"""Presence scores and expression-transfer evaluation on the MI355X backend.

Mirror of ``EvaluationMixin.estimate_presence_score`` / ``process_presence_scores`` of the reference
(``model/evaluate.py:426-521``): the raw score of a reference cell is the column sum of the un-normalised Gaussian
query->reference connectivity matrix, overall and per query group; the O(n_query * k) accumulation runs in one HIP
kernel (``cm_presence_scores``), the O(n_reference) post-processing (log1p, percentile clipping, min-max) stays on the
host like upstream.

``evaluate_expression_transfer`` mirrors ``evaluate.py:236-323``: the per-gene agreement between measured and imputed
expression (Pearson r, z-scored RMSE, Jensen-Shannon distance) is one column-moment kernel family
(``cm_expression_metrics``) instead of a Python loop over genes calling scipy; Spearman rho ranks every gene column
on the device (``rank_kernels.hip``).  The label
evaluation metrics of the reference's mixin (sklearn classification reports, plots) are out of scope.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from scipy.sparse import issparse

from . import _native
from .logging import logger


def process_presence_scores(scores: pd.DataFrame, log: bool = False, percentile: tuple[float, float] = (1, 99)) -> pd.DataFrame:
    """log1p (optional) -> percentile clipping per column (unless (0, 100)) -> min-max normalisation per column."""
    out = scores.copy()
    if log:
        out = np.log1p(out)
    cols = {}
    for name in out.columns:
        x = out[name].to_numpy(copy=True)
        if tuple(percentile) != (0, 100):
            lo, hi = np.percentile(x, percentile[0]), np.percentile(x, percentile[1])
            x = np.clip(x, lo, hi)
        mn, mx = np.min(x), np.max(x)
        cols[name] = (x - mn) / (mx - mn) if mx > mn else np.zeros_like(x)
    return pd.DataFrame(cols, index=out.index, columns=out.columns)


class PresenceMixin:
    """Adds ``estimate_presence_score`` to :class:`cellmapper_amd.CellMapper`."""

    def estimate_presence_score(self, groupby: str | None = None, key_added: str = "presence_score", log: bool = False,
                                percentile: tuple[float, float] = (1, 99)):
        """Presence score of every reference cell: written to ``reference.obs[key_added]`` (all query cells) and, with
        ``groupby``, per group to ``reference.obsm[key_added]`` (DataFrame, one column per group in order of appearance)."""
        if self.knn is None or self.knn.yx is None:
            raise ValueError("Neighbors must be computed before estimating presence scores.")
        yx = self.knn.yx
        names = self.reference.obs_names
        groups, codes = None, None
        if groupby is not None:
            labels = self.query.obs[groupby]
            groups = list(pd.unique(labels))
            lookup = {g: i for i, g in enumerate(groups)}
            codes = np.fromiter((lookup[v] for v in labels), dtype=np.int32, count=len(labels))
        raw = _native.default_handle().presence_scores(yx.distances, yx.indices, yx.shape[1], codes,
                                                       len(groups) if groups else 0)
        df_all = process_presence_scores(pd.DataFrame({"all": raw[:, 0]}, index=names), log=log, percentile=percentile)
        self.reference.obs[key_added] = df_all["all"]
        logger.info("Presence score across all query cells computed and stored in `reference.obs['%s']`", key_added)
        if groupby is not None:
            df = pd.DataFrame(raw[:, 1:].astype(np.float32), index=names, columns=groups)
            self.reference.obsm[key_added] = process_presence_scores(df, log=log, percentile=percentile)
            logger.info("Presence scores per group defined in `query.obs['%s']` computed and stored in "
                        "`reference.obsm['%s']`", groupby, key_added)


class ExpressionEvaluationMixin:
    """Adds ``evaluate_expression_transfer`` to :class:`cellmapper_amd.CellMapper` (reference evaluate.py:236-323)."""

    def _get_aligned_expression_arrays(self, layer_key: str):
        """Dense (imputed, original) arrays restricted to the genes shared by ``query_imputed`` and ``query``
        (reference evaluate.py:325-352)."""
        if self.query_imputed is None:
            raise ValueError("Imputed query data not found. Either run map_layers() first or set query_imputed manually.")
        shared = list(self.query_imputed.var_names.intersection(self.query.var_names))
        if len(shared) == 0:
            raise ValueError("No shared genes between query_imputed and query.")
        ii = self.query_imputed.var_names.get_indexer(shared)
        iq = self.query.var_names.get_indexer(shared)
        imputed = self.query_imputed.X
        original = self.query.X if layer_key == "X" else self.query.layers[layer_key]
        imputed = imputed[:, ii]
        original = original[:, iq]
        if issparse(imputed):
            imputed = imputed.toarray()
        if issparse(original):
            original = original.toarray()
        return np.asarray(imputed), np.asarray(original), shared

    def _store_expression_metric(self, shared_genes, values, method, test_var_key=None) -> None:
        """Per-gene values -> ``query.var``, validity mask, average over (test) genes (reference evaluate.py:354-424)."""
        var = self.query.var
        var[f"metric_{method}"] = np.nan
        var.loc[shared_genes, f"metric_{method}"] = values
        valid = ~np.isnan(values)
        var[f"_is_valid_test_gene_{method}"] = False
        var.loc[shared_genes, f"_is_valid_test_gene_{method}"] = valid
        n_test = int(np.sum(valid))
        if test_var_key is not None:
            test_mask = var[test_var_key].astype(bool)
            var[f"_is_valid_test_gene_{method}"] = var[f"_is_valid_test_gene_{method}"] & test_mask
            n_test = int(var[f"_is_valid_test_gene_{method}"].sum())
            if n_test == 0:
                raise ValueError(f"No valid test genes found using '{test_var_key}'")
        vals = var.loc[var[f"_is_valid_test_gene_{method}"], f"metric_{method}"]
        avg = float(np.mean(vals))
        self.expression_transfer_metrics = {"method": method, "average": avg, "n_shared_genes": len(shared_genes),
                                            "n_test_genes": n_test}
        logger.info("Expression transfer evaluation (%s): average value = %.4f (n_shared_genes=%d, n_test_genes=%d)",
                    method, avg, len(shared_genes), n_test)

    def evaluate_expression_transfer(self, layer_key: str = "X", method: str = "pearson", groupby: str | None = None,
                                     test_var_key: str | None = None) -> None:
        """Agreement between imputed and measured expression per gene, overall (``query.var['metric_<method>']``) and,
        with ``groupby``, per group of query cells (``query.varm['metric_<method>']``, genes x groups in order of
        appearance).  ``expression_transfer_metrics`` holds the average over the valid (test) genes."""
        if method not in ("pearson", "spearman", "js", "jensen-shannon", "rmse"):
            raise NotImplementedError(f"Method '{method}' is not implemented.")
        imputed, original, shared = self._get_aligned_expression_arrays(layer_key)
        groups, codes = None, None
        if groupby is not None:
            labels = self.query.obs[groupby]
            groups = list(pd.unique(labels))
            lookup = {g: i for i, g in enumerate(groups)}
            codes = np.fromiter((lookup[v] for v in labels), dtype=np.int32, count=len(labels))
        res = _native.default_handle().expression_metrics(original, imputed, method, codes, len(groups) if groups else 0)
        self._store_expression_metric(shared, res[0], method, test_var_key)
        if groupby is not None:
            df = pd.DataFrame(np.full((self.query.n_vars, len(groups)), np.nan, dtype=np.float32),
                              index=self.query.var_names, columns=groups)
            df.loc[shared, :] = res[1:].T
            self.query.varm[f"metric_{method}"] = df
            logger.info("Metrics per group defined in `query.obs['%s']` computed and stored in `query.varm['%s']`",
                        groupby, f"metric_{method}")
