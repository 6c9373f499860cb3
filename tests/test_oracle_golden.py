"""Pin the CPU oracle (oracle/cm_oracle.c + oracle/oracle.py) against golden vectors produced by running the
reference itself (oracle/make_golden.py; sklearn 1.7.2 / scipy 1.15.3 / numpy 2.2.6).  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from parity import (RTOL_FLOAT, assert_csr_close, assert_knn_parity, golden_csr, golden_names, load_golden, tie_rows)

MAPPING_CASES = [n for n in golden_names() if n.startswith(("synth_", "spatial", "identity", "bin", "latlon"))]
COUNT_METRICS = ("hamming", "jaccard", "dice", "rogerstanimoto", "russellrao", "sokalmichener", "sokalsneath", "yule")
KERNELS = ["gaussian", "scarches", "inverse_distance", "equal"]


def rep(g):
    x, y = g["xrep"], g["yrep"]
    n = min(50, x.shape[1])  # get_n_comps, cellmapper.py:246-248 / utils.py:223-227
    return np.ascontiguousarray(x[:, :n]), np.ascontiguousarray(y[:, :n])


@pytest.mark.parametrize("name", MAPPING_CASES)
def test_oracle_matches_reference(name, oracle):
    g = load_golden(name)
    metric = str(g["metric"])
    k = int(g["k"])
    xrep, yrep = rep(g)
    if metric == "cosine":
        dist, idx = oracle.knn_cosine(xrep, yrep, k)
    elif metric in ("manhattan", "chebyshev", "canberra", "braycurtis"):
        dist, idx = oracle.knn_minkowski_family(xrep, yrep, k, metric)
        if metric in ("canberra", "braycurtis"):
            assert np.array_equal(dist, g["yx_dist"])   # distances bit for bit (neighbours: tie groups, below)
    elif metric in COUNT_METRICS:
        # ratios of integer counts: the distances bit for bit; equal distances are ordered arbitrarily upstream (argpartition under
        # brute force, heap order on the ball tree) and discrete values tie in masses, also across the k-th position
        dist, idx = oracle.knn_count_family(xrep, yrep, k, metric)
        assert np.array_equal(dist, g["yx_dist"], equal_nan=True)
    elif metric == "correlation":
        dist, idx = oracle.knn_correlation(xrep, yrep, k)
    elif metric == "haversine":
        dist, idx = oracle.knn_haversine(xrep, yrep, k)
    elif metric == "nan_euclidean":
        dist, idx = oracle.knn_nan_euclidean(xrep, yrep, k)
    elif metric == "sqeuclidean":
        dist, idx = oracle.knn_euclidean(xrep, yrep, k, squared=True)
        # the float64 GEMM of sklearn's chunks accumulates in BLAS order (OpenBLAS kernels), the oracle in a sequential fma chain:
        # last-bit differences in the squared values (1e-15 relative), identical neighbours.  ("euclidean" hides them behind the
        # square root for float64 input.)
        np.testing.assert_allclose(dist, g["yx_dist"], rtol=2e-15, atol=0)
        assert np.array_equal(idx, g["yx_idx"])
    else:
        dist, idx = oracle.knn_euclidean(xrep, yrep, k)
    assert dist.dtype == g["yx_dist"].dtype
    n_tie = assert_knn_parity(dist, idx, g["yx_dist"], g["yx_idx"], xrep, yrep, metric)
    trows = tie_rows(idx, g["yx_idx"])
    if "dups" not in name and metric not in COUNT_METRICS and metric not in ("haversine", "nan_euclidean"):
        assert len(trows) == 0, f"{n_tie} rows differ only by tie order, {len(trows)} by tie membership"
    # feed the GOLDEN graph into the downstream stages so each stage is pinned independently
    gd, gi = g["yx_dist"], g["yx_idx"]  # float32 for sklearn cosine on float32 input
    n_ref = xrep.shape[0]
    for kern in KERNELS:
        w, _ = oracle.kernel_weights(gd, gi, kern)
        indptr, cols, _, v64 = oracle.mapping_matrix(w, gd, gi, n_ref, normalize=False)
        assert_csr_close(indptr, cols, v64, golden_csr(g, f"conn_{kern}"), rtol=1e-12)
    for kern in KERNELS:
        if f"map_{kern}_data" not in g:
            continue
        w, _ = oracle.kernel_weights(gd, gi, kern)
        indptr, cols, v32, _ = oracle.mapping_matrix(w, gd, gi, n_ref)
        gm = golden_csr(g, f"map_{kern}")
        assert gm.dtype == np.float32
        assert_csr_close(indptr, cols, v32, gm, rtol=2e-7)
        # downstream stages use the GOLDEN mapping matrix
        gm.sort_indices()
        ip, cc, vv = gm.indptr.astype(np.int64), gm.indices.astype(np.int32), gm.data
        code, conf = oracle.label_vote(ip, cc, vv, g["label_codes"], len(g["categories"]))
        cat_code = np.array([int(s[1:]) for s in g["categories"]])
        assert np.array_equal(cat_code[code], g[f"pred_code_{kern}"])
        assert np.array_equal(conf, g[f"conf_{kern}"])  # float32 sums in scipy's order: bit-exact
        num = oracle.spmm_dense(ip, cc, vv, g["num_values"])
        assert num.dtype == g[f"num_pred_{kern}"].dtype
        np.testing.assert_array_equal(num, g[f"num_pred_{kern}"])
        om = oracle.spmm_dense(ip, cc, vv, g["obsm_in"])
        assert om.dtype == g[f"obsm_pred_{kern}"].dtype
        np.testing.assert_array_equal(om, g[f"obsm_pred_{kern}"])
        ld = oracle.spmm_dense(ip, cc, vv, g["X_dense"])
        np.testing.assert_array_equal(ld, g[f"layer_dense_{kern}"])
        ls = oracle.spmm_csr(ip, cc, vv, golden_csr(g, "X_sparse"))
        gs = golden_csr(g, f"layer_sparse_{kern}")
        assert gs.dtype == np.float32
        np.testing.assert_array_equal(ls, gs.toarray())
        assert sp.csr_matrix(ls).nnz == gs.nnz  # csr_matmat drops exact zeros


def test_oracle_small_data(oracle):
    """tests/conftest.py:23-28 of the reference: 5 points in 2-D, k=3 (contains exact ties)."""
    g = load_golden("small_data")
    for graph in ("xx", "yy", "xy", "yx"):
        dist, idx = oracle.knn_euclidean(g["xrep"], g["yrep"], 3)
        assert_knn_parity(dist, idx, g[f"{graph}_dist"], g[f"{graph}_idx"], g["xrep"], g["yrep"])
    gd, gi = g["yx_dist"], g["yx_idx"]
    for kern in KERNELS:
        w, _ = oracle.kernel_weights(gd, gi, kern)
        indptr, cols, _, v64 = oracle.mapping_matrix(w, gd, gi, 5, normalize=False)
        assert_csr_close(indptr, cols, v64, golden_csr(g, f"conn_{kern}"), rtol=1e-14)


def test_oracle_sample_graph(oracle):
    """tests/conftest.py:11-20 of the reference through every kernel + normalisation."""
    g = load_golden("sample_graph")
    gd, gi = g["distances"], g["indices"]
    for kern in KERNELS:
        w, _ = oracle.kernel_weights(gd, gi, kern)
        indptr, cols, v32, v64 = oracle.mapping_matrix(w, gd, gi, 3)
        assert_csr_close(indptr, cols, v64, golden_csr(g, f"conn_{kern}"), rtol=1e-14)
        assert_csr_close(indptr, cols, v32, golden_csr(g, f"map_{kern}"), rtol=2e-7)
    w, _ = oracle.kernel_weights(gd, gi, "inverse_distance", epsilon=0.5)
    indptr, cols, _, v64 = oracle.mapping_matrix(w, gd, gi, 3, normalize=False)
    assert_csr_close(indptr, cols, v64, golden_csr(g, "conn_inverse_distance_eps"), rtol=1e-14)


@pytest.mark.parametrize("tag", ["none", "true", "false"])
def test_oracle_padded_graph(tag, oracle):
    """Padded (-1 / inf) neighbour lists as produced by Neighbors.from_distances (knn.py:296-337)."""
    g = load_golden("padded_graph")
    gd, gi = g[f"dist_{tag}"], g[f"idx_{tag}"]
    n = gi.shape[0]
    for kern in KERNELS:
        w, _ = oracle.kernel_weights(gd, gi, kern)
        indptr, cols, v32, v64 = oracle.mapping_matrix(w, gd, gi, n)
        assert_csr_close(indptr, cols, v64, golden_csr(g, f"conn_{kern}_{tag}"), rtol=1e-13)
        assert_csr_close(indptr, cols, v32, golden_csr(g, f"map_{kern}_{tag}"), rtol=2e-7)


def test_oracle_errors(oracle):
    with pytest.raises(ValueError):
        oracle.kernel_weights(np.full((2, 2), np.inf), np.full((2, 2), -1), "gaussian")
    with pytest.raises(ValueError):
        oracle.knn_euclidean(np.zeros((3, 20), np.float32), np.zeros((2, 20), np.float32), 4)


@pytest.mark.parametrize("inc,tag", [(None, "none"), (True, "true"), (False, "false")])
def test_oracle_precomputed_distance_ingest_matches_reference(oracle, inc, tag):
    """oracle.extract_neighbors_from_distances vs the padded arrays the reference produced (utils.py:129-219)."""
    g = load_golden("padded_graph")
    idx, dist = oracle.extract_neighbors_from_distances(golden_csr(g, "dm"), include_self=inc)
    assert idx.dtype == np.int64 and dist.dtype == np.float64
    assert np.array_equal(dist, g[f"dist_{tag}"])
    assert np.array_equal(idx, g[f"idx_{tag}"])


def test_oracle_precomputed_distance_ingest_edge_cases(oracle):
    import scipy.sparse as sp

    idx, dist = oracle.extract_neighbors_from_distances(sp.csr_matrix((4, 4)))
    assert idx.shape == (4, 0) and dist.shape == (4, 0)
    m = sp.csr_matrix(np.array([[0, 2.0, 1.0], [np.inf, 0, 0], [0, 0, 0]]))
    idx, dist = oracle.extract_neighbors_from_distances(m, include_self=True)
    assert idx[0].tolist() == [0, 2, 1] and dist[0].tolist() == [0.0, 1.0, 2.0]
    assert np.isinf(dist[1]).sum() >= 1


@pytest.mark.parametrize("method", ["pearson", "rmse", "js", "spearman"])
def test_oracle_expression_metrics_match_reference(oracle, method):
    """oracle.expression_metrics vs the values the reference's helpers produced (evaluate.py:20-63,271-289), overall and
    per group; genes with a constant / all-zero / all-negative column exercise the NaN and std -> 1 rules."""
    g = load_golden("expr_eval")
    r = oracle.expression_metrics(g["orig"], g["imputed"], method)
    assert np.allclose(r, g[f"{method}_all"], rtol=1e-5, atol=2e-6, equal_nan=True)
    assert np.array_equal(np.isnan(r), np.isnan(g[f"{method}_all"]))
    for j, name in enumerate(g["group_names"]):
        r = oracle.expression_metrics(g["orig"], g["imputed"], method, g["labels"] == name)
        assert np.allclose(r, g[f"{method}_groups"][:, j], rtol=1e-5, atol=2e-6, equal_nan=True)



def test_joint_embedding_oracle_matches_reference_golden(oracle):
    """Fast CCA (utils.py:230-398, the reference's own ARPACK run) and joint PCA (sklearn's arpack PCA = scanpy.pp.pca's
    default path) against the dense LAPACK restatement: singular values 1e-7, every singular vector up to its sign 1e-6."""
    import scipy.sparse as sp

    g = load_golden("joint_embedding")
    nc = int(g["n_comps"])
    Sq = sp.csr_matrix((g["Sq_data"], g["Sq_indices"], g["Sq_indptr"]), shape=tuple(g["Sq_shape"]))
    Sr = sp.csr_matrix((g["Sr_data"], g["Sr_indices"], g["Sr_indptr"]), shape=tuple(g["Sr_shape"]))
    for tag, X, Y, zc in (("dense", g["Xq"], g["Xr"], True), ("nocenter", g["Xq"], g["Xr"], False), ("sparse", Sq, Sr, True)):
        U, s, Vt = oracle.cross_cov_svd(X, Y, nc, zero_center=zc)
        np.testing.assert_allclose(s, g[f"{tag}_s"], rtol=1e-7)
        assert np.max(np.abs(np.abs(np.sum(U * g[f"{tag}_U"], axis=0)) - 1)) < 1e-6
        assert np.max(np.abs(np.abs(np.sum(Vt * g[f"{tag}_Vt"], axis=1)) - 1)) < 1e-6
        assert oracle.subspace_error(U, g[f"{tag}_U"]) < 1e-5
    P = oracle.joint_pca(np.concatenate([g["Xr"], g["Xq"]]), nc)
    np.testing.assert_allclose(P, g["pca_scores"], atol=2e-5 * np.abs(P).max())
