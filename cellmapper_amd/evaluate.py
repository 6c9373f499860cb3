"""Presence scores on the MI355X backend.

Mirror of ``EvaluationMixin.estimate_presence_score`` / ``process_presence_scores`` of the reference
(``model/evaluate.py:426-521``): the raw score of a reference cell is the column sum of the un-normalised Gaussian
query->reference connectivity matrix, overall and per query group; the O(n_query * k) accumulation runs in one HIP
kernel (``cm_presence_scores``), the O(n_reference) post-processing (log1p, percentile clipping, min-max) stays on the
host like upstream.  The label / expression evaluation metrics of the reference's mixin are out of scope.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

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
