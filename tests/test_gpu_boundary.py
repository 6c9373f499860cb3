"""GPU tests of the drop-in boundary: CellMapper / Neighbors / NeighborsResults used exactly like the reference's
tests use them (tests/model/test_neighbors*.py, test_query_to_reference_mapping.py, test_self_mapping.py),
compared with golden vectors from the reference."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp
from scipy.sparse import csr_matrix

from parity import assert_csr_close, assert_knn_parity, golden_csr, load_golden

pytestmark = pytest.mark.gpu


def build_pair(g, rep_key="X_emb", sparse_x=False):
    from cellmapper_amd import MiniAnnData

    n_ref, nq = g["xrep"].shape[0], g["yrep"].shape[0]
    X = golden_csr(g, "X_sparse") if sparse_x else g["X_dense"]
    ref = MiniAnnData(X=X, obs=pd.DataFrame({"label": pd.Categorical([f"c{c:02d}" for c in g["label_codes"]]),
                                             "num": g["num_values"]}, index=[f"r{i}" for i in range(n_ref)]),
                      obsm={rep_key: g["xrep"], "X_umap": g["obsm_in"]},
                      uns={"label_colors": [f"#{i:06x}" for i in range(len(g["categories"]))]})
    qry = MiniAnnData(X=np.zeros((nq, 3), np.float32), obs=pd.DataFrame(index=[f"q{i}" for i in range(nq)]),
                      obsm={rep_key: g["yrep"]})
    return qry, ref


@pytest.mark.parametrize("name,kern", [("synth_f32_k30", "gaussian"), ("synth_f32_k30", "scarches"),
                                       ("synth_f32_k15", "inverse_distance"), ("synth_f32_k30", "equal"),
                                       ("synth_f64_k30", "gaussian"), ("synth_clustered_k30", "gaussian"),
                                       ("synth_d64_trunc50_k15", "gaussian"), ("synth_d10_k15", "gaussian"),
                                       ("synth_f32_manhattan_k15", "gaussian"), ("synth_f64_chebyshev_k15", "gaussian")])
def test_map_matches_reference(name, kern):
    from cellmapper_amd import CellMapper

    g = load_golden(name)
    qry, ref = build_pair(g)
    cm = CellMapper(qry, ref)
    out = cm.map(obs_keys=["label", "num"], obsm_keys="X_umap", layer_key="X", n_neighbors=int(g["k"]),
                 use_rep="X_emb", knn_method="sklearn", metric=str(g["metric"]), only_yx=True, mapping_method=kern)
    assert out is cm and cm.only_yx is True and cm.prediction_postfix == "pred" and cm.confidence_postfix == "conf"
    n = min(50, g["xrep"].shape[1])
    assert_knn_parity(cm.knn.yx.distances, cm.knn.yx.indices, g["yx_dist"], g["yx_idx"], g["xrep"][:, :n],
                      g["yrep"][:, :n], str(g["metric"]))
    assert cm.knn.yx.n_targets == ref.n_obs and cm.knn.xx is None
    mm = cm.mapping_matrix
    assert isinstance(mm, csr_matrix) and mm.dtype == np.float32 and mm.shape == (qry.n_obs, ref.n_obs)
    assert_csr_close(mm.indptr, mm.indices, mm.data, golden_csr(g, f"map_{kern}"), rtol=1e-5)
    pred = qry.obs["label_pred"]
    assert isinstance(pred.dtype, pd.CategoricalDtype)
    got = np.array([int(s[1:]) for s in pred.astype(str)])
    assert (got == g[f"pred_code_{kern}"]).mean() > 0.995
    np.testing.assert_allclose(qry.obs["label_conf"].to_numpy(), g[f"conf_{kern}"], rtol=1e-5)
    assert qry.obs["label_conf"].dtype == np.float32
    assert "num_conf" not in qry.obs.columns and qry.obs["num_pred"].dtype == np.float64
    np.testing.assert_allclose(qry.obs["num_pred"].to_numpy(), g[f"num_pred_{kern}"], rtol=1e-5, atol=1e-7)
    assert qry.obsm["X_umap_pred"].shape == (qry.n_obs, 2) and qry.obsm["X_umap_pred"].dtype == np.float32
    np.testing.assert_allclose(qry.obsm["X_umap_pred"], g[f"obsm_pred_{kern}"], rtol=1e-5, atol=1e-6)
    qi = cm.query_imputed
    assert qi.X.shape == (qry.n_obs, ref.n_vars) and qi.obs is qry.obs and qi.var is ref.var
    np.testing.assert_allclose(qi.X, g[f"layer_dense_{kern}"], rtol=1e-5, atol=1e-6)
    assert "label_pred_colors" in qry.uns


@pytest.mark.parametrize("name", ["bin40_jaccard_k15", "bin10_hamming_k10", "bin40_yule_k15"])
def test_counting_metrics_through_the_class(name):
    """`metric=` is handed through (cellmapper.py:251 -> knn.py:429): a boolean metric warns that the data is read as `x != 0`
    (sklearn's DataConversionWarning is a UserWarning), hamming does not; distances equal the reference's bit for bit, neighbours
    up to the ties upstream orders arbitrarily."""
    import warnings
    from cellmapper_amd import CellMapper

    g = load_golden(name)
    qry, ref = build_pair(g)
    metric = str(g["metric"])
    cm = CellMapper(qry, ref)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        cm.compute_neighbors(n_neighbors=int(g["k"]), use_rep="X_emb", metric=metric, only_yx=True)
    said = any("converted to boolean" in str(x.message) for x in w)
    assert said == (metric != "hamming")
    assert np.array_equal(cm.knn.yx.distances, g["yx_dist"], equal_nan=True)
    n = min(50, g["xrep"].shape[1])
    assert_knn_parity(cm.knn.yx.distances, cm.knn.yx.indices, g["yx_dist"], g["yx_idx"], g["xrep"][:, :n], g["yrep"][:, :n], metric)
    cm.compute_mapping_matrix(method="gaussian")
    cm.map_obs(key="label")
    assert qry.obs["label_pred"].notna().all()


def test_sparse_layer_and_full_mode_and_jaccard():
    from cellmapper_amd import CellMapper

    g = load_golden("synth_full_d37_k10")
    qry, ref = build_pair(g, sparse_x=True)
    cm = CellMapper(qry, ref)
    cm.compute_neighbors(n_neighbors=10, use_rep="X_emb", only_yx=False)
    for gname in ("xx", "yy", "xy", "yx"):
        res = getattr(cm.knn, gname)
        a = g["xrep"] if gname in ("xx", "yx") else g["yrep"]
        b = g["xrep"] if gname in ("xx", "xy") else g["yrep"]
        assert_knn_parity(res.distances, res.indices, g[f"{gname}_dist"], g[f"{gname}_idx"], a, b)
    assert cm.knn.xy.n_targets == qry.n_obs and cm.knn.xx.n_targets == ref.n_obs
    cm.compute_mapping_matrix("gaussian")
    cm.map_layers("X")
    res = cm.query_imputed.X
    assert sp.issparse(res) and res.dtype == np.float32
    gs = golden_csr(g, "layer_sparse_gaussian")
    np.testing.assert_allclose(res.toarray(), gs.toarray(), rtol=1e-5, atol=1e-7)
    for meth in ("jaccard", "hnoca"):
        cm.compute_mapping_matrix(meth)
        mm = cm.mapping_matrix
        assert_csr_close(mm.indptr, mm.indices, mm.data, golden_csr(g, f"map_{meth}"), rtol=1e-5)
    cm.only_yx = True
    with pytest.raises(ValueError, match="require both x and y"):
        cm.compute_mapping_matrix("jaccard")


def test_identity_mapping_k1():
    """Reference tests/model/test_query_to_reference_mapping.py:75-92: k=1 + jaccard reproduces labels exactly."""
    from cellmapper_amd import CellMapper

    g = load_golden("identity_k1")
    qry, ref = build_pair(g)
    cm = CellMapper(qry, ref)
    cm.map(obs_keys=["label", "num"], use_rep="X_emb", n_neighbors=1, mapping_method="jaccard")
    assert (qry.obs["label_pred"].astype(str).to_numpy() == ref.obs["label"].astype(str).to_numpy()).all()
    np.testing.assert_allclose(qry.obs["num_pred"].to_numpy(), g["num_values"], rtol=1e-6)
    mm = cm.mapping_matrix
    assert_csr_close(mm.indptr, mm.indices, mm.data, golden_csr(g, "map_jaccard"), rtol=1e-6)


def test_self_mapping_spatial():
    from cellmapper_amd import CellMapper, MiniAnnData

    g = load_golden("spatial32_self_k20")
    n = g["xrep"].shape[0]
    ad = MiniAnnData(X=g["X_dense"], obs=pd.DataFrame({"label": pd.Categorical([f"c{c:02d}" for c in g["label_codes"]])}),
                     obsm={"spatial": g["xrep"]})
    cm = CellMapper(ad)
    assert cm.reference is ad and cm.query is ad
    cm.map(obs_keys="label", layer_key="X", use_rep="spatial", n_neighbors=20, only_yx=True)
    assert_knn_parity(cm.knn.yx.distances, cm.knn.yx.indices, g["yx_dist"], g["yx_idx"], g["xrep"], g["yrep"], atol=1e-9)
    assert (cm.knn.yx.indices[:, 0] == np.arange(n)).all() and (cm.knn.yx.distances[:, 0] == 0).all()
    np.testing.assert_allclose(cm.query_imputed.X, g["layer_dense_gaussian"], rtol=1e-5, atol=1e-6)


def test_neighbors_results_graph_api():
    """Reference tests/model/test_neighbors_results.py, on the GPU path."""
    from cellmapper_amd import Neighbors, NeighborsResults

    g = load_golden("small_data")
    nb = Neighbors(g["xrep"], g["yrep"])
    nb.compute_neighbors(n_neighbors=3, method="sklearn")
    assert "xx=True" in repr(nb) and "yx=True" in repr(nb)
    for graph in ("xx", "yy", "xy", "yx"):
        r = getattr(nb, graph)
        assert_knn_parity(r.distances, r.indices, g[f"{graph}_dist"], g[f"{graph}_idx"], g["xrep"], g["yrep"])
    adj = nb.get_adjacency_matrices()
    assert all(isinstance(a, csr_matrix) and (a.data == 1).all() for a in adj)
    gold = NeighborsResults(g["yx_dist"], g["yx_idx"])
    np.testing.assert_array_equal(gold.boolean_adjacency().toarray(), golden_csr(g, "bool_adj").toarray())
    np.testing.assert_allclose(gold.knn_graph_distances.toarray(), golden_csr(g, "graph_dist").toarray())
    assert np.allclose(gold.knn_graph_distances.diagonal(), 0)
    for kern in ("gaussian", "scarches", "random", "inverse_distance", "equal"):
        m = gold.knn_graph_connectivities(kernel=kern)
        assert isinstance(m, csr_matrix) and m.shape == (5, 5) and m.dtype == np.float64
        if kern != "random":
            assert (m.data > 0).all()
            np.testing.assert_allclose(m.toarray(), golden_csr(g, f"conn_{kern}").toarray(), rtol=1e-13)
    assert gold.boolean_adjacency(set_diag=False).diagonal().sum() == 0
    assert gold.boolean_adjacency(set_diag=True).diagonal().sum() == 5
    with pytest.raises(ValueError, match="square"):
        NeighborsResults(g["yx_dist"], g["yx_idx"], n_targets=9).boolean_adjacency(set_diag=True)


def test_from_distances_and_precomputed():
    from cellmapper_amd import CellMapper, MiniAnnData, Neighbors

    g = load_golden("padded_graph")
    dm = golden_csr(g, "dm")
    for inc, tag in ((None, "none"), (True, "true"), (False, "false")):
        nb = Neighbors.from_distances(dm, include_self=inc)
        assert nb._is_self_mapping and nb.xx is nb.yy is nb.xy is nb.yx
        conn = nb.xx.knn_graph_connectivities()
        np.testing.assert_allclose(conn.toarray(), golden_csr(g, f"conn_gaussian_{tag}").toarray(), rtol=1e-13)
    n = dm.shape[0]
    ad = MiniAnnData(X=np.zeros((n, 2), np.float32), obs=pd.DataFrame({"v": np.arange(n, dtype=float)}), obsp={"distances": dm})
    cm = CellMapper(ad)
    cm.load_precomputed_distances(include_self=True)
    cm.compute_mapping_matrix("gaussian")
    mm = cm.mapping_matrix
    assert_csr_close(mm.indptr, mm.indices, mm.data, golden_csr(g, "map_gaussian_true"), rtol=2e-7)
    cm.map_obs("v")
    assert ad.obs["v_pred"].shape == (n,)


def test_mapping_matrix_setter_and_errors():
    from cellmapper_amd import CellMapper

    g = load_golden("synth_f32_k15")
    qry, ref = build_pair(g)
    cm = CellMapper(qry, ref)
    with pytest.raises(ValueError, match="shape mismatch"):
        cm.mapping_matrix = sp.identity(5, format="csr")
    raw = golden_csr(g, "conn_gaussian")
    cm.mapping_matrix = raw.tocoo()
    mm = cm.mapping_matrix
    assert mm.dtype == np.float32
    assert_csr_close(mm.indptr, mm.indices, mm.data, golden_csr(g, "map_gaussian"), rtol=2e-7)
    np.testing.assert_allclose(np.asarray(mm.sum(1)).ravel(), 1.0, rtol=1e-5)
    with pytest.raises(KeyError):
        cm.map_obs("missing")
    with pytest.raises(ValueError, match="n_neighbors <= n_samples_fit"):
        cm.compute_neighbors(n_neighbors=10**6, use_rep="X_emb")
    with pytest.raises(ValueError, match="not supported"):
        cm.compute_neighbors(n_neighbors=5, use_rep="X_emb", metric="mahalanobis")
    cm.map(use_rep="X_emb", n_neighbors=5)  # no keys: warning only


def test_cosine_through_boundary():
    from cellmapper_amd import CellMapper

    g = load_golden("synth_f32_cosine_k15")
    qry, ref = build_pair(g)
    cm = CellMapper(qry, ref)
    cm.map(obs_keys="label", use_rep="X_emb", n_neighbors=15, metric="cosine", only_yx=True)
    assert cm.knn.yx.distances.dtype == np.float32
    assert_knn_parity(cm.knn.yx.distances, cm.knn.yx.indices, g["yx_dist"], g["yx_idx"], g["xrep"], g["yrep"], "cosine")
    mm = cm.mapping_matrix
    assert_csr_close(mm.indptr, mm.indices, mm.data, golden_csr(g, "map_gaussian"), rtol=1e-5)


def test_sharded_search_equals_single_search():
    """Reference rows split into shards (as the multi-GPU driver does), per-shard top-k with global indices,
    k-way merge: must reproduce the single-search result exactly (SURVEY.md 8e tie policy)."""
    import torch
    from cellmapper_amd.device import DevicePipeline
    from cellmapper_amd.sharded import shard_bounds

    rng = np.random.default_rng(3)
    n_ref, nq, d, k = 9001, 1200, 50, 30
    ref = rng.standard_normal((n_ref, d)).astype(np.float32)
    ref[500] = ref[8000]  # a cross-shard exact tie
    qry = rng.standard_normal((nq, d)).astype(np.float32)
    qry[0] = ref[500]
    pipe = DevicePipeline(0)
    with torch.cuda.stream(pipe.stream):
        tr, tq = torch.from_numpy(ref).cuda(), torch.from_numpy(qry).cuda()
        d_full, i_full = pipe.knn(tr, tq, k)
        parts_d, parts_i = [], []
        for r in range(3):
            lo, hi = shard_bounds(n_ref, 3, r)
            dd, ii = pipe.knn(tr[lo:hi].contiguous(), tq, k, idx_offset=lo, squared=True)
            parts_d.append(dd), parts_i.append(ii)
        md, mi = pipe.merge(torch.stack(parts_d), torch.stack(parts_i), k)
        torch.cuda.synchronize()
    assert torch.equal(mi, i_full) and torch.equal(md, d_full)
    assert set(mi[0, :2].tolist()) == {500, 8000} and mi[0, 0].item() == 500


def test_presence_scores_match_reference():
    """estimate_presence_score (reference evaluate.py:426-521) vs golden output of the reference."""
    from cellmapper_amd import CellMapper, MiniAnnData

    g = load_golden("presence")
    n_ref, nq = g["xrep"].shape[0], g["yrep"].shape[0]
    for tag, kw in (("default", {}), ("log", dict(log=True)), ("nopct", dict(percentile=(0, 100)))):
        ref = MiniAnnData(X=np.zeros((n_ref, 2), np.float32), obs=pd.DataFrame(index=[f"r{i}" for i in range(n_ref)]),
                          obsm={"X_emb": g["xrep"]})
        qry = MiniAnnData(X=np.zeros((nq, 2), np.float32), obs=pd.DataFrame({"grp": g["groups"]}, index=[f"q{i}" for i in range(nq)]),
                          obsm={"X_emb": g["yrep"]})
        cm = CellMapper(qry, ref)
        with pytest.raises(ValueError, match="Neighbors must be computed"):
            cm.estimate_presence_score()
        cm.compute_neighbors(n_neighbors=int(g["k"]), use_rep="X_emb", only_yx=True)
        cm.estimate_presence_score(groupby="grp", key_added="ps", **kw)
        np.testing.assert_allclose(ref.obs["ps"].to_numpy(), g[f"all_{tag}"], rtol=1e-6, atol=1e-9)
        df = ref.obsm["ps"]
        assert [str(c) for c in df.columns] == [str(c) for c in g["group_names"]]
        np.testing.assert_allclose(df.to_numpy(), g[f"groups_{tag}"], rtol=2e-5, atol=1e-6)
    from cellmapper_amd import _native
    raw = _native.default_handle().presence_scores(g["yx_dist"], g["yx_idx"], n_ref)
    np.testing.assert_allclose(raw[:, 0], g["raw_all"], rtol=1e-12)


class _SoloDist:
    """torch.distributed stand-in for a single process (the collectives of the sharded driver are skipped at world 1)."""

    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def get_rank():
        return 0


@pytest.mark.parametrize("kernel", ["gaussian", "scarches", "inverse_distance"])
def test_sharded_driver_device_ops_match_fused_pipeline(kernel):
    """The device-tier ops the multi-GPU driver composes (squared local k-NN, merge, moments, bandwidth, weights,
    ELL->CSR, vote) must reproduce the fused single-GPU pipeline bit for bit."""
    import torch
    from cellmapper_amd.device import DevicePipeline
    from cellmapper_amd.sharded import HipOps, shard_bounds, sharded_map_labels

    rng = np.random.default_rng(12)
    n_ref, nq, d, k, n_cat = 20000, 3000, 50, 30, 20
    pipe = DevicePipeline(0)
    with torch.cuda.stream(pipe.stream):
        ref = torch.from_numpy(rng.standard_normal((n_ref, d)).astype(np.float32)).cuda()
        qry = torch.from_numpy(rng.standard_normal((nq, d)).astype(np.float32)).cuda()
        codes = torch.from_numpy(rng.integers(0, n_cat, n_ref).astype(np.int32)).cuda()
        code0, conf0, d0, i0 = pipe.map_labels(ref, qry, k, codes, n_cat, kernel=kernel, want_graph=True)
        ops = HipOps(pipe)
        code1, conf1, (lo, hi), d1, i1 = sharded_map_labels(ops, _SoloDist, ref, 0, n_ref, qry, k, codes, n_cat, kernel=kernel)
        # emulate two reference shards + the merge by hand, then the per-slice tail of the driver
        parts = [pipe.knn(ref[a:b].contiguous(), qry, k, idx_offset=a, squared=True)
                 for a, b in (shard_bounds(n_ref, 2, r) for r in range(2))]
        dm, im = pipe.merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)
        torch.cuda.synchronize()
    assert (lo, hi) == (0, nq)
    assert torch.equal(i1, i0) and torch.equal(d1, d0) and torch.equal(im, i0) and torch.equal(dm, d0)
    assert torch.equal(code1, code0) and torch.equal(conf1, conf0)


@pytest.mark.parametrize("inc,tag", [(None, "none"), (True, "true"), (False, "false")])
def test_precomputed_distance_ingest_matches_reference(inc, tag):
    """cm_ingest_distances (through cellmapper_amd.utils) vs the reference's padded arrays (utils.py:129-219)."""
    from cellmapper_amd.utils import extract_neighbors_from_distances

    g = load_golden("padded_graph")
    idx, dist = extract_neighbors_from_distances(golden_csr(g, "dm"), include_self=inc)
    assert idx.dtype == np.int64 and dist.dtype == np.float64
    assert np.array_equal(dist, g[f"dist_{tag}"])
    assert np.array_equal(idx, g[f"idx_{tag}"])


def test_precomputed_distance_ingest_random_graphs_and_edges(oracle):
    """Ragged rows, unsorted rows with ties, empty rows, explicit diagonal entries, a long row: GPU == oracle."""
    from cellmapper_amd.utils import extract_neighbors_from_distances

    rng = np.random.default_rng(41)
    n = 3000
    lens = rng.integers(0, 40, n)
    lens[7] = 900                                   # a long row for the wave-per-row kernel (up to 1 023 entries)
    lens[11] = 0
    rows = np.repeat(np.arange(n), lens)
    cols = np.concatenate([rng.choice(n, l, replace=False) for l in lens]) if lens.sum() else np.zeros(0, int)
    vals = np.round(rng.random(len(rows)) * 8) / 4  # many ties
    sel = rng.random(n) < 0.5                       # half of the rows stored sorted by distance
    m = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    m.sort_indices()
    for i in np.nonzero(sel)[0]:
        lo, hi = m.indptr[i], m.indptr[i + 1]
        o = np.argsort(m.data[lo:hi], kind="stable")
        m.data[lo:hi], m.indices[lo:hi] = m.data[lo:hi][o], m.indices[lo:hi][o]
    m.has_sorted_indices = False
    for inc in (None, True, False):
        gi, gd = extract_neighbors_from_distances(m, include_self=inc)
        oi, od = oracle.extract_neighbors_from_distances(m, include_self=inc)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od), inc
    gi, gd = extract_neighbors_from_distances(sp.csr_matrix((4, 4)))
    assert gi.shape == (4, 0) and gd.shape == (4, 0)
    # rows with 1 024+ stored neighbours: one workgroup per row, the row in LDS (up to 12 287 entries)
    n2 = 5000
    lens2 = rng.integers(0, 30, n2)
    lens2[3], lens2[4], lens2[77] = 1500, 4200, 1024
    rows2 = np.repeat(np.arange(n2), lens2)
    cols2 = np.concatenate([rng.choice(n2, l, replace=False) for l in lens2])
    m2 = sp.csr_matrix((np.round(rng.random(len(rows2)) * 64) / 8, (rows2, cols2)), shape=(n2, n2))
    m2.sort_indices()
    lo, hi = m2.indptr[4], m2.indptr[5]           # one of the long rows stored sorted by distance
    o = np.argsort(m2.data[lo:hi], kind="stable")
    m2.data[lo:hi], m2.indices[lo:hi] = m2.data[lo:hi][o], m2.indices[lo:hi][o]
    m2.has_sorted_indices = False
    for inc in (None, True, False):
        gi, gd = extract_neighbors_from_distances(m2, include_self=inc)
        oi, od = oracle.extract_neighbors_from_distances(m2, include_self=inc)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od), inc
    # rows beyond the in-LDS sorts (round 4: no row limit any more -- staged stable radix sort): two rows of 12 400 / 12 900
    # stored neighbours with many tied distances, one of them stored sorted, next to short rows; both include_self rules
    n3 = 13000
    lens3 = rng.integers(0, 6, n3)
    lens3[5], lens3[9000] = 12400, 12900
    rows3 = np.repeat(np.arange(n3), lens3)
    cols3 = np.concatenate([rng.choice(n3, l, replace=False) for l in lens3])
    m3 = sp.csr_matrix((np.round(rng.random(len(rows3)) * 512) / 64, (rows3, cols3)), shape=(n3, n3))
    m3.sort_indices()
    lo, hi = m3.indptr[9000], m3.indptr[9001]
    o = np.argsort(m3.data[lo:hi], kind="stable")
    m3.data[lo:hi], m3.indices[lo:hi] = m3.data[lo:hi][o], m3.indices[lo:hi][o]
    m3.has_sorted_indices = False
    for inc in (None, True, False):
        gi, gd = extract_neighbors_from_distances(m3, include_self=inc)
        oi, od = oracle.extract_neighbors_from_distances(m3, include_self=inc)
        assert gi.shape[1] >= 12890 and np.array_equal(gi, oi) and np.array_equal(gd, od), inc


@pytest.mark.parametrize("method", ["pearson", "rmse", "js", "spearman"])
def test_expression_metrics_match_reference(oracle, method):
    """cm_expression_metrics vs the reference's per-gene values (golden) and, on a larger ragged case, vs the oracle."""
    from cellmapper_amd import _native

    hip = _native.default_handle()
    g = load_golden("expr_eval")
    groups = list(g["group_names"])
    codes = np.array([groups.index(v) for v in g["labels"]], np.int32)
    res = hip.expression_metrics(g["orig"], g["imputed"], method, codes, len(groups))
    assert res.shape == (len(groups) + 1, g["orig"].shape[1]) and res.dtype == np.float32
    assert np.allclose(res[0], g[f"{method}_all"], rtol=1e-5, atol=2e-6, equal_nan=True)
    assert np.array_equal(np.isnan(res[0]), np.isnan(g[f"{method}_all"]))
    assert np.allclose(res[1:].T, g[f"{method}_groups"], rtol=1e-5, atol=2e-6, equal_nan=True)
    rng = np.random.default_rng(5)
    n, G = 20011, 333                               # several row chunks, a partial last column block
    a = rng.poisson(rng.gamma(2.0, 2.0, G), (n, G)).astype(np.float32)
    b = (0.5 * a + rng.standard_normal((n, G))).astype(np.float32)
    a[:, 17] = 3.0
    b[:, 40] = 0.0
    codes = rng.integers(0, 4, n).astype(np.int32)
    res = hip.expression_metrics(a, b, method, codes, 5)   # group 4 is empty: NaN row
    assert np.allclose(res[0], oracle.expression_metrics(a, b, method), rtol=1e-5, atol=2e-6, equal_nan=True)
    assert np.allclose(res[2], oracle.expression_metrics(a, b, method, codes == 1), rtol=1e-5, atol=2e-6, equal_nan=True)
    assert np.isnan(res[5]).all()


def test_evaluate_expression_transfer_side_effects():
    """Drop-in surface of evaluate_expression_transfer (evaluate.py:236-323): query.var / query.varm columns, summary
    dict, test-gene mask, errors."""
    from cellmapper_amd import CellMapper, MiniAnnData

    g = load_golden("expr_eval")
    n, G = g["orig"].shape
    genes = [f"g{i}" for i in range(G)]
    query = MiniAnnData(X=g["orig"], obs=pd.DataFrame({"grp": g["labels"]}, index=[f"q{i}" for i in range(n)]),
                        var=pd.DataFrame({"is_test": np.arange(G) % 2 == 0}, index=genes), obsm={"X_emb": np.zeros((n, 2), np.float32)})
    ref = MiniAnnData(X=np.zeros((5, G), np.float32), var=pd.DataFrame(index=genes), obsm={"X_emb": np.zeros((5, 2), np.float32)})
    cm = CellMapper(query, ref)
    with pytest.raises(ValueError, match="Imputed query data not found"):
        cm.evaluate_expression_transfer()
    cm.query_imputed = MiniAnnData(X=g["imputed"], obs=query.obs.copy(), var=pd.DataFrame(index=genes))
    cm.evaluate_expression_transfer(method="pearson", groupby="grp", test_var_key="is_test")
    got = query.var["metric_pearson"].to_numpy()
    assert np.allclose(got, g["pearson_all"], rtol=1e-5, atol=2e-6, equal_nan=True)
    df = query.varm["metric_pearson"]
    assert list(df.columns) == list(g["group_names"])
    assert np.allclose(df.to_numpy(), g["pearson_groups"], rtol=1e-5, atol=2e-6, equal_nan=True)
    valid = ~np.isnan(g["pearson_all"]) & (np.arange(G) % 2 == 0)
    m = cm.expression_transfer_metrics
    assert m["method"] == "pearson" and m["n_shared_genes"] == G and m["n_test_genes"] == int(valid.sum())
    assert abs(m["average"] - float(np.mean(g["pearson_all"][valid]))) < 1e-5
    cm.evaluate_expression_transfer(method="spearman")
    assert np.allclose(query.var["metric_spearman"].to_numpy(), g["spearman_all"], rtol=1e-5, atol=2e-6, equal_nan=True)
    with pytest.raises(NotImplementedError):
        cm.evaluate_expression_transfer(method="kendall")



def test_staged_flow_keeps_results_on_the_device(oracle):
    """compute_neighbors -> compute_mapping_matrix -> map_obs / map_obsm / map_layers through the graph tier (cm_graph_*):
    nothing is copied back until the caller reads it, and what is read equals the host-tier results and the oracle."""
    import pandas as pd
    import scipy.sparse as sp

    from cellmapper_amd import CellMapper, MiniAnnData, _native

    rng = np.random.default_rng(5)
    n_ref, nq, d, k, G = 6000, 1500, 40, 20, 96
    ref = rng.standard_normal((n_ref, d)).astype(np.float32)
    qry = rng.standard_normal((nq, d)).astype(np.float32)
    codes = rng.integers(0, 9, n_ref)
    umap = rng.standard_normal((n_ref, 2)).astype(np.float32)
    X = sp.random(n_ref, G, density=0.1, format="csr", dtype=np.float32, random_state=rng)
    Xd = rng.random((n_ref, G), dtype=np.float32)
    reference = MiniAnnData(X=X, obs=pd.DataFrame({"label": pd.Categorical([f"c{c}" for c in codes]),
                                                   "age": rng.random(n_ref)}),
                            obsm={"X_emb": ref, "X_umap": umap}, layers={"dense": Xd})
    query = MiniAnnData(X=np.zeros((nq, 3), np.float32), obsm={"X_emb": qry})
    cm = CellMapper(query, reference)
    cm.compute_neighbors(n_neighbors=k, use_rep="X_emb", only_yx=True)
    yx = cm.knn.yx
    assert yx._distances is None and yx.shape == (nq, n_ref) and yx.n_neighbors == k   # still on the device
    cm.compute_mapping_matrix("gaussian")
    cm.map_obs("label")
    cm.map_obs("age")
    cm.map_obsm("X_umap")
    cm.map_layers("X")
    sparse_imputed = cm.query_imputed.X
    cm.map_layers("dense")
    assert yx._distances is None and cm._mapping_matrix is None, "the staged flow must not have materialised host copies"
    o = oracle.run_path(ref, qry, k, labels=np.unique([f"c{c}" for c in codes], return_inverse=True)[1].astype(np.int32), n_cat=9,
                        numeric=np.asarray(reference.obs["age"]), obsm=umap, layer=Xd)
    assert np.array_equal(yx.indices, o["idx"]) and np.array_equal(yx.distances, o["dist"])
    m = cm.mapping_matrix
    assert np.array_equal(m.indices, o["cols"]) and m.dtype == np.float32
    np.testing.assert_allclose(m.data, o["vals"], rtol=1e-5)
    h = _native.default_handle()
    ip = m.indptr.astype(np.int64)
    code, conf = h.label_vote(ip, m.indices, m.data, np.unique([f"c{c}" for c in codes], return_inverse=True)[1].astype(np.int32), 9)
    np.testing.assert_array_equal(query.obs["label_conf"].to_numpy(), conf)           # graph tier == host tier, bit for bit
    np.testing.assert_array_equal(query.obsm["X_umap_pred"], h.spmm_dense(ip, m.indices, m.data, umap))
    np.testing.assert_array_equal(cm.query_imputed.X, h.spmm_dense(ip, m.indices, m.data, Xd))
    np.testing.assert_array_equal(query.obs["age_pred"].to_numpy(), h.spmm_dense(ip, m.indices, m.data, np.asarray(reference.obs["age"])))
    want = sp.csr_matrix(oracle.spmm_csr(ip, m.indices, m.data, X))
    want.sort_indices()
    assert np.array_equal(sparse_imputed.indices, want.indices) and np.array_equal(sparse_imputed.data, want.data)
    # a caller-supplied mapping matrix replaces the resident one (host-tier path from then on)
    cm.mapping_matrix = m
    assert cm._mapping_graph is None
    cm.map_obsm("X_umap")
    np.testing.assert_array_equal(query.obsm["X_umap_pred"], h.spmm_dense(ip, m.indices, m.data, umap))


def test_joint_representations_match_reference_and_oracle(oracle):
    """fast CCA / joint PCA through cm_gram + cm_project (float64 matrix cores) against the
    reference's own truncated_svd_cross_covariance outputs (golden) and the dense LAPACK oracle; then the `use_rep=None`
    flow of CellMapper.map (cellmapper.py:211-235) end to end.  Tolerances: singular values 1e-7 relative, singular vectors
    1e-6 up to their sign (the reference's ARPACK tolerance; the float32 inputs are centred in float64 here and in
    float32 upstream), PCA scores 2e-5 of their range."""
    import pandas as pd
    import scipy.sparse as sp

    from cellmapper_amd import CellMapper, MiniAnnData, _native
    from cellmapper_amd.embedding import joint_pca_scores, truncated_svd_cross_covariance
    from parity import load_golden

    g = load_golden("joint_embedding")
    nc = int(g["n_comps"])
    h = _native.default_handle()
    # primitives against numpy
    Xq64 = g["Xq"].astype(np.float64)
    rowc = Xq64 - Xq64.mean(axis=1)[:, None]
    np.testing.assert_allclose(h.gram(g["Xq"], center_rows=True), rowc.T @ rowc, rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(h.gram(g["Xq"]), Xq64.T @ Xq64, rtol=1e-12)
    P = np.random.default_rng(0).standard_normal((g["Xq"].shape[1], 70))   # > 64 directions: two passes
    np.testing.assert_allclose(h.project(g["Xq"], P, center_rows=True), rowc @ P, rtol=1e-11, atol=1e-10)
    Sq = sp.csr_matrix((g["Sq_data"], g["Sq_indices"], g["Sq_indptr"]), shape=tuple(g["Sq_shape"]))
    Sr = sp.csr_matrix((g["Sr_data"], g["Sr_indices"], g["Sr_indptr"]), shape=tuple(g["Sr_shape"]))
    Sd = np.asarray(Sq.todense(), np.float64)
    np.testing.assert_allclose(h.gram(Sq), Sd.T @ Sd, rtol=1e-12)
    mu = Sd.mean(axis=0)
    np.testing.assert_allclose(h.project(Sq, P[:, :5], col_shift=mu), (Sd - mu) @ P[:, :5], rtol=1e-11, atol=1e-10)
    for tag, X, Y, zc in (("dense", g["Xq"], g["Xr"], True), ("nocenter", g["Xq"], g["Xr"], False), ("sparse", Sq, Sr, True)):
        U, s, Vt = truncated_svd_cross_covariance(X, Y, n_comps=nc, zero_center=zc)
        np.testing.assert_allclose(s, g[f"{tag}_s"], rtol=1e-7)
        assert np.max(np.abs(np.abs(np.sum(U * g[f"{tag}_U"], axis=0)) - 1)) < 1e-6
        assert np.max(np.abs(np.abs(np.sum(Vt * g[f"{tag}_Vt"], axis=1)) - 1)) < 1e-6
        Uo, so, _ = oracle.cross_cov_svd(X, Y, nc, zero_center=zc)
        np.testing.assert_allclose(s, so, rtol=1e-9)
        assert oracle.subspace_error(U, Uo) < 1e-7
    Z = joint_pca_scores(np.concatenate([g["Xr"], g["Xq"]]), nc)
    np.testing.assert_allclose(Z, g["pca_scores"], atol=2e-5 * np.abs(Z).max())
    np.testing.assert_allclose(Z, oracle.joint_pca(np.concatenate([g["Xr"], g["Xq"]]), nc), atol=1e-9 * np.abs(Z).max())
    # the drop-in flow: use_rep=None -> fast CCA on shared genes -> neighbours -> label transfer
    nq, nr = g["Xq"].shape[0], g["Xr"].shape[0]
    genes_r = [f"g{i}" for i in range(g["Xr"].shape[1])]
    genes_q = genes_r[::-1]                                              # same genes, different order: inner join by name
    labels = pd.Categorical([f"t{i % 4}" for i in range(nr)])
    reference = MiniAnnData(X=g["Xr"], obs=pd.DataFrame({"label": labels}), var=pd.DataFrame(index=genes_r))
    query = MiniAnnData(X=np.ascontiguousarray(g["Xq"][:, ::-1]), var=pd.DataFrame(index=genes_q))
    cm = CellMapper(query, reference).map(obs_keys="label", n_neighbors=8, only_yx=True)
    assert query.obsm["X_cca"].shape == (nq, nc) or query.obsm["X_cca"].shape[1] == min(50, g["Xq"].shape[1])
    assert reference.uns["X_cca_params"]["n_common_genes"] == g["Xq"].shape[1]
    np.testing.assert_allclose(np.linalg.norm(query.obsm["X_cca"], axis=1), 1.0, rtol=1e-12)    # l2_scale
    assert "label_pred" in query.obs and cm.knn.yx.n_neighbors == 8
    cm2 = CellMapper(query, reference)
    cm2.compute_joint_pca(n_comps=nc)
    np.testing.assert_allclose(reference.obsm["X_pca"], g["pca_scores"][:nr], atol=2e-5 * np.abs(Z).max())
    np.testing.assert_allclose(query.obsm["X_pca"], g["pca_scores"][nr:], atol=2e-5 * np.abs(Z).max())


def test_shared_device_graph_is_not_silently_remapped(oracle):
    """`cm.knn.yx` and `cm` share ONE device graph.  Re-running the mapping on it through NeighborsResults with another
    kernel must not change what the CellMapper applies (round-2 advisor finding): the mapper restores its own matrix."""
    import pandas as pd

    from cellmapper_amd import CellMapper, MiniAnnData

    rng = np.random.default_rng(9)
    n_ref, nq, d, k = 4000, 900, 30, 12
    ref = rng.standard_normal((n_ref, d)).astype(np.float32)
    qry = rng.standard_normal((nq, d)).astype(np.float32)
    umap = rng.standard_normal((n_ref, 2)).astype(np.float32)
    reference = MiniAnnData(X=np.zeros((n_ref, 2), np.float32), obs=pd.DataFrame(index=[str(i) for i in range(n_ref)]),
                            obsm={"X_emb": ref, "X_umap": umap})
    query = MiniAnnData(X=np.zeros((nq, 2), np.float32), obsm={"X_emb": qry})
    cm = CellMapper(query, reference)
    cm.compute_neighbors(n_neighbors=k, use_rep="X_emb", only_yx=True)
    cm.compute_mapping_matrix("gaussian")
    before = cm.mapping_matrix.copy()
    other = cm.knn.yx.normalized_mapping_matrix("equal")          # re-maps the SHARED graph in place
    assert np.allclose(other.data, 1.0 / k)
    cm.map_obsm("X_umap")
    o = oracle.run_path(ref, qry, k, obsm=umap)
    np.testing.assert_allclose(query.obsm["X_umap_pred"], o["obsm_pred"], rtol=1e-5, atol=1e-6)   # gaussian weights, not equal
    after = cm.mapping_matrix
    assert np.array_equal(after.indices, before.indices) and np.array_equal(after.data, before.data)


# ---- round 4: NaN / infinity screen on the device copies, pinned staging ring of the host tier -------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_non_finite_input_raises_value_error_like_sklearn(dtype, bad):
    """sklearn's check_array rejects NaN / infinity before any search (knn.py:429-440); here the screen runs on the
    uploaded copy (cm_rep_create / cm_knn), first, last (sub-16-byte tail) and middle elements, either side."""
    from cellmapper_amd import Neighbors, _native

    rng = np.random.default_rng(5)
    ref = rng.standard_normal((301, 7)).astype(dtype)   # 301 * 7 elements: not a multiple of 16 bytes in float32
    qry = rng.standard_normal((33, 7)).astype(dtype)
    h = _native.default_handle()
    d0, i0 = h.knn(ref, qry, 5)
    assert np.isfinite(d0).all()
    for side in ("ref", "qry"):
        for pos in ((0, 0), (-1, -1), (17, 3)):
            r, q = ref.copy(), qry.copy()
            (r if side == "ref" else q)[pos] = bad
            with pytest.raises(ValueError, match="NaN or infinity"):
                h.knn(r, q, 5)
            with pytest.raises(ValueError, match="NaN or infinity"):
                Neighbors(r, q).compute_neighbors(n_neighbors=5, only_yx=True)
    d1, i1 = h.knn(ref, qry, 5)   # the handle keeps working after a rejected call
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)


def test_staging_ring_moves_the_same_bytes(monkeypatch, oracle):
    """Opt-in (CELLMAP_XFER=1) large host <-> device copies go through the pinned staging ring (cm_api.hip: xfer_run); forced here for small
    arrays with several chunks per thread, odd sizes and both directions: results equal the plain-copy path bit for bit."""
    from cellmapper_amd import _native

    rng = np.random.default_rng(11)
    ref = rng.standard_normal((20011, 50)).astype(np.float32)
    qry = rng.standard_normal((3001, 50)).astype(np.float32)
    codes = rng.integers(0, 9, ref.shape[0]).astype(np.int32)
    B = rng.random((ref.shape[0], 37), dtype=np.float32)
    outs = []
    for ring in (False, True):
        if ring:
            monkeypatch.setenv("CELLMAP_XFER", "1")
            monkeypatch.setenv("CELLMAP_XFER_MIN_KB", "1"); monkeypatch.setenv("CELLMAP_XFER_SLOT_KB", "96")
            monkeypatch.setenv("CELLMAP_XFER_THREADS", "3")
        else:
            monkeypatch.setenv("CELLMAP_XFER", "0")
        _native.reload_knobs()
        h = _native.Handle(0)
        xr, yr = _native.DeviceRep(h, ref), _native.DeviceRep(h, qry)
        g = _native.DeviceGraph(h, xr, yr, 15)
        dist, idx = g.fetch_knn()
        g.mapping("gaussian")
        ip, cc, vv = g.fetch_mapping()
        code, conf = g.label_vote(codes, 9)
        lay = g.spmm_dense(B)
        d2, i2 = h.knn(ref, qry, 15)
        outs.append((dist, idx, ip, cc, vv, code, conf, lay, d2, i2))
        g.close(); xr.close(); yr.close(); h.close()
        monkeypatch.delenv("CELLMAP_XFER", raising=False)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    o = oracle.knn_euclidean(ref, qry, 15)
    assert np.array_equal(outs[1][1], o[1]) and np.array_equal(outs[1][0], o[0])
    _native.reload_knobs()


def test_large_downloads_take_the_ring_by_default(monkeypatch):
    """Round 5: without any knob, a download of 64 MB or more (k-NN lists, a dense layer product) goes through the staging ring,
    created on first use; smaller results never create it.  Same bytes as the plain path (CELLMAP_XFER=0)."""
    from cellmapper_amd import _native

    rng = np.random.default_rng(12)
    ref = rng.standard_normal((30_000, 20)).astype(np.float32)
    qry = rng.standard_normal((120_000, 20)).astype(np.float32)
    B = rng.random((ref.shape[0], 160), dtype=np.float32)
    codes = rng.integers(0, 9, ref.shape[0]).astype(np.int32)
    outs, pinned = [], []
    for plain in (True, False):
        if plain:
            monkeypatch.setenv("CELLMAP_XFER", "0")
        else:
            monkeypatch.delenv("CELLMAP_XFER", raising=False)
        _native.reload_knobs()
        h = _native.Handle(0)
        xr, yr = _native.DeviceRep(h, ref), _native.DeviceRep(h, qry)
        g = _native.DeviceGraph(h, xr, yr, 70)
        g.mapping("gaussian")
        code, conf = g.label_vote(codes, 9)             # 480 KB + 480 KB down: plain path
        small_only = h.alloc_stats()["pinned_calls"]
        dist, idx = g.fetch_knn()                        # 2 x 67 MB
        lay = g.spmm_dense(B)                            # 77 MB, in query tiles
        pinned.append((small_only, h.alloc_stats()["pinned_calls"]))
        outs.append((code, conf, dist, idx, lay))
        g.close(); xr.close(); yr.close(); h.close()
    monkeypatch.delenv("CELLMAP_XFER", raising=False)
    _native.reload_knobs()
    assert pinned == [(0, 0), (0, 1)], pinned
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_default_map_searches_yx_at_once_and_the_other_graphs_on_first_access(oracle):
    """`CellMapper.map()` defaults to `only_yx=False` upstream: four searches (knn.py:441-457), three of which nothing reads unless
    jaccard / hnoca / `get_adjacency_matrices` is used.  Here `yx` is searched at once, `xx` / `yy` / `xy` on first access: the
    same arrays (oracle self-joins), the same early error when n_neighbors exceeds the smaller set."""
    import pandas as pd

    from cellmapper_amd import CellMapper, MiniAnnData
    from cellmapper_amd.knn import Neighbors

    rng = np.random.default_rng(21)
    ref = rng.standard_normal((3000, 20)).astype(np.float32)
    qry = rng.standard_normal((400, 20)).astype(np.float32)
    labels = pd.Categorical([f"t{c}" for c in rng.integers(0, 5, ref.shape[0])])
    R = MiniAnnData(X=np.zeros((ref.shape[0], 1), np.float32), obs=pd.DataFrame({"label": labels}), obsm={"X_emb": ref})
    Q = MiniAnnData(X=np.zeros((qry.shape[0], 1), np.float32), obsm={"X_emb": qry})
    cm = CellMapper(Q, R).map(obs_keys="label", use_rep="X_emb", n_neighbors=10)      # only_yx=False, the default
    nb = cm.knn
    assert nb._res["yx"] is not None and nb._res["xx"] is None and nb._res["yy"] is None and nb._res["xy"] is None
    assert "xx=True" in repr(nb) and nb._res["xx"] is None                              # (asking does not compute)
    for name, (a, b) in {"xx": (ref, ref), "yy": (qry, qry), "xy": (qry, ref), "yx": (ref, qry)}.items():
        od, oi = oracle.knn_euclidean(a, b, 10)
        res = getattr(nb, name)
        assert np.array_equal(res.indices, oi) and np.array_equal(res.distances, od), name
    assert all(nb._res[n] is not None for n in ("xx", "yy", "xy", "yx"))
    xx, yy, xy, yx = nb.get_adjacency_matrices()
    assert xx.shape == (3000, 3000) and yy.shape == (400, 400) and xy.shape == (3000, 400) and yx.shape == (400, 3000)
    cm2 = CellMapper(Q, R)
    cm2.compute_neighbors(n_neighbors=10, use_rep="X_emb")
    cm2.compute_mapping_matrix("jaccard")                                              # reads all four: computed on demand
    assert cm2.mapping_matrix.shape == (400, 3000)
    with pytest.raises(ValueError, match="n_neighbors <= n_samples_fit"):
        Neighbors(ref, qry).compute_neighbors(n_neighbors=500)                          # yy would need 500 <= 400
    nb3 = Neighbors(ref, qry)
    nb3.compute_neighbors(n_neighbors=500, only_yx=True)                                # yx alone is fine: 500 <= 3000
    assert nb3.yx.indices.shape == (400, 500) and nb3.xx is None
