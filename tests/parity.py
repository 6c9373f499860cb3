"""Shared parity helpers: tie-aware k-NN comparison and golden-fixture loading."""
from __future__ import annotations

import glob
import os

import numpy as np
import scipy.sparse as sp

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerances (BASELINE.json north_star): neighbour indices bit-exact for exact brute force
# (tie groups compared as sets), transferred floats within 1e-5 relative.
RTOL_FLOAT = 1e-5
RTOL_DIST = 1e-6


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_csr(g, prefix):
    shape = tuple(int(v) for v in g[f"{prefix}_shape"])
    return sp.csr_matrix((g[f"{prefix}_data"], g[f"{prefix}_indices"], g[f"{prefix}_indptr"]), shape=shape)


def exact_dist(ref, qry_row, cols, metric):
    r = ref[cols].astype(np.float64)
    q = qry_row.astype(np.float64)
    if metric == "cosine":
        rn = r / np.maximum(np.linalg.norm(r, axis=1, keepdims=True), 1e-300)
        qn = q / max(np.linalg.norm(q), 1e-300)
        return np.clip(1.0 - rn @ qn, 0.0, 2.0)
    if metric == "correlation":
        rc = r - r.mean(axis=1, keepdims=True)
        qc = q - q.mean()
        return 1.0 - np.clip((rc @ qc) / np.maximum(np.linalg.norm(rc, axis=1) * np.linalg.norm(qc), 1e-300), -1.0, 1.0)
    if metric == "manhattan":
        return np.abs(r - q).sum(1)
    if metric == "chebyshev":
        return np.abs(r - q).max(1)
    if metric == "canberra":
        den = np.abs(r) + np.abs(q)
        return np.divide(np.abs(r - q), den, out=np.zeros_like(den), where=den > 0).sum(1)
    if metric == "braycurtis":
        den = (np.abs(r) + np.abs(q)).sum(1)
        return np.divide(np.abs(r - q).sum(1), den, out=np.zeros_like(den), where=den > 0)
    if metric == "sqeuclidean":
        return ((r - q) ** 2).sum(1)
    if metric == "haversine":
        s0, s1 = np.sin(0.5 * (q[0] - r[:, 0])), np.sin(0.5 * (q[1] - r[:, 1]))
        return 2.0 * np.arcsin(np.sqrt(s0 * s0 + np.cos(q[0]) * np.cos(r[:, 0]) * s1 * s1))
    if metric == "nan_euclidean":
        both = ~np.isnan(r) & ~np.isnan(q)[None, :]
        df = np.where(both, r - q[None, :], 0.0)
        cnt = both.sum(1)
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(cnt > 0, np.sqrt((df * df).sum(1) / np.maximum(cnt, 1) * r.shape[1]), np.nan)
    if metric == "hamming":
        return (r != q).sum(1) / float(r.shape[1])
    if metric in ("jaccard", "dice", "rogerstanimoto", "russellrao", "sokalmichener", "sokalsneath", "yule"):
        rb, qb = r != 0, q != 0
        n = float(r.shape[1])
        ntt = (rb & qb).sum(1).astype(np.float64); ntf = (qb & ~rb).sum(1).astype(np.float64); nft = (~qb & rb).sum(1).astype(np.float64)
        R = ntf + nft; nff = n - ntt - R
        with np.errstate(divide="ignore", invalid="ignore"):
            if metric == "jaccard":
                return np.divide(R, ntt + R, out=np.zeros_like(R), where=(ntt + R) > 0)
            if metric == "dice":
                return R / (2.0 * ntt + R)
            if metric == "russellrao":
                return (n - ntt) / n
            if metric == "sokalsneath":
                return (2.0 * R) / (ntt + 2.0 * R)
            if metric == "yule":
                return np.divide(2.0 * ntf * nft, ntt * nff + ntf * nft, out=np.zeros_like(R), where=(ntf * nft) > 0)
            return (2.0 * R) / (ntt + nff + 2.0 * R)
    return np.sqrt(((r - q) ** 2).sum(1))


def assert_knn_parity(dist, idx, g_dist, g_idx, ref, qry, metric="euclidean", rtol=RTOL_DIST, atol=1e-5,
                      tie_tol=1e-9):
    """Indices must be identical, except inside groups of (numerically) tied distances, which are compared
    as sets; a tie group touching the k-th position may hold any index whose true distance equals the tie value."""
    dist = np.asarray(dist)
    idx = np.asarray(idx)
    assert dist.shape == g_dist.shape and idx.shape == g_idx.shape, (dist.shape, g_dist.shape)
    if metric == "cosine":
        # sklearn computes cosine distances in the input dtype (float32 stays float32): rounding-level noise
        atol = max(atol, 4e-7 if g_dist.dtype == np.float32 else 1e-12)
        tie_tol = max(tie_tol, 4e-7 if g_dist.dtype == np.float32 else 1e-12)
    if metric == "correlation":
        # scipy's C loop, numpy's pairwise mean and this path's float64 arithmetic round differently in the last bits of
        # 1 - cos (values in [0, 2]): 1e-13 absolute on the distances, neighbours equal except inside ties at that level
        rtol, atol, tie_tol = 0.0, 1e-13, 1e-13
    if metric == "haversine":
        # libm / numpy / device sin, cos, asin differ in the last bit; float32 data under brute force is ranked by a float32-rounded
        # bracket (HaversineDistance32.rdist): 1e-15 (float64 arithmetic) / 3e-7 (float32-rounded brackets) on angles in [0, pi]
        f32 = np.asarray(ref).dtype == np.float32
        rtol, atol, tie_tol = (3e-7, 3e-7, 6e-7) if f32 else (4e-15, 1e-15, 1e-14)
    if metric == "nan_euclidean":
        # upstream: GEMM-form distances of zero-filled rows minus BLAS correction products -- for float32 data the corrections and the
        # subtraction run in float32, a cancellation that leaves up to 4e-5 relative noise in the reference's OWN values (measured
        # on the d = 10 golden, 15 % missing); here: direct float64 differences, rounded once.  NaN (no shared coordinate) must
        # match exactly, as NaN.
        f32 = np.asarray(g_dist).dtype == np.float32
        rtol, atol, tie_tol = (1e-4, 1e-6, 2e-4) if f32 else (1e-12, 1e-13, 1e-11)
        assert np.array_equal(np.isnan(dist), np.isnan(g_dist))
        nanrow = np.isnan(g_dist)
        dist, g_dist = np.where(nanrow, 0.0, dist), np.where(nanrow, 0.0, g_dist)
        idx, g_idx = np.where(nanrow, -1, idx), np.where(nanrow, -1, g_idx)   # (which rows fill a NaN tail is arbitrary upstream)
    np.testing.assert_allclose(dist, g_dist, rtol=rtol, atol=atol)
    bad_rows = np.nonzero((idx != g_idx).any(axis=1))[0]
    k = idx.shape[1]
    for i in bad_rows:
        gd = g_dist[i].astype(np.float64)
        for pos in np.nonzero(idx[i] != g_idx[i])[0]:
            tol = tie_tol * max(1.0, abs(gd[pos])) if metric not in ("cosine", "correlation") else tie_tol
            grp = np.nonzero(np.abs(gd - gd[pos]) <= tol)[0]
            ours, theirs = set(idx[i, grp].tolist()), set(g_idx[i, grp].tolist())
            if ours == theirs:
                continue
            assert grp[-1] == k - 1, f"row {i}: index sets differ inside a non-boundary tie group at pos {pos}"
            extra = np.array(sorted(ours - theirs), dtype=np.int64)
            de = exact_dist(ref, qry[i], extra, metric)
            assert np.all(np.abs(de - gd[pos]) <= max(tol, atol)), (
                f"row {i}: boundary-tie indices {extra} have distances {de}, expected {gd[pos]}")
    return len(bad_rows)


def assert_csr_close(indptr, cols, vals, g_csr, rtol=RTOL_FLOAT, atol=0.0, tie_rows_ok=None):
    """Compare a CSR triple with a golden scipy CSR (canonical form)."""
    g = g_csr.tocsr().copy()
    g.sum_duplicates()
    g.sort_indices()
    ours = sp.csr_matrix((np.asarray(vals), np.asarray(cols), np.asarray(indptr)), shape=g.shape)
    ours.sort_indices()
    if tie_rows_ok is None or len(tie_rows_ok) == 0:
        assert np.array_equal(ours.indptr, g.indptr)
        assert np.array_equal(ours.indices, g.indices)
        np.testing.assert_allclose(ours.data, g.data, rtol=rtol, atol=atol)
    else:
        keep = np.setdiff1d(np.arange(g.shape[0]), np.asarray(tie_rows_ok))
        a, b = ours[keep], g[keep]
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
        np.testing.assert_allclose(a.data, b.data, rtol=rtol, atol=atol)


def tie_rows(idx, g_idx):
    return np.nonzero((np.sort(np.asarray(idx), 1) != np.sort(np.asarray(g_idx), 1)).any(axis=1))[0]
