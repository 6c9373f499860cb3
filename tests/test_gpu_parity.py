"""GPU parity tests: the HIP path (through the C ABI, ctypes) against golden vectors from the reference and
against the CPU oracle on seeded inputs.  Run on the MI355X box with `pytest -m gpu`."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from parity import (assert_csr_close, assert_knn_parity, golden_csr, golden_names, load_golden, tie_rows)

pytestmark = pytest.mark.gpu

MAPPING_CASES = [n for n in golden_names() if n.startswith(("synth_", "spatial", "identity"))]
KERNELS = ["gaussian", "scarches", "inverse_distance", "equal"]


@pytest.fixture(scope="module")
def hip():
    from cellmapper_amd import _native

    return _native.default_handle()


def rep(g):
    x, y = g["xrep"], g["yrep"]
    n = min(50, x.shape[1])
    return np.ascontiguousarray(x[:, :n]), np.ascontiguousarray(y[:, :n])


@pytest.mark.parametrize("name", MAPPING_CASES)
def test_knn_matches_reference_golden(name, hip):
    g = load_golden(name)
    metric, k = str(g["metric"]), int(g["k"])
    xrep, yrep = rep(g)
    dist, idx = hip.knn(xrep, yrep, k, metric=metric)
    assert dist.dtype == g["yx_dist"].dtype and idx.dtype == np.int64
    assert_knn_parity(dist, idx, g["yx_dist"], g["yx_idx"], xrep, yrep, metric)
    if "dups" not in name:
        ok_rows = ~np.isnan(g["yx_dist"]).any(axis=1)   # (nan_euclidean: which rows fill an all-NaN list is arbitrary upstream)
        assert len(tie_rows(idx[ok_rows], g["yx_idx"][ok_rows])) == 0


@pytest.mark.parametrize("name", MAPPING_CASES)
def test_graph_and_transfer_match_reference_golden(name, hip):
    g = load_golden(name)
    gd, gi = g["yx_dist"].astype(np.float64), g["yx_idx"]
    n_ref = g["xrep"].shape[0]
    f32_dist = g["yx_dist"].dtype == np.float32  # reference keeps float32 through the kernel stage (cosine)
    for kern in KERNELS:
        indptr, cols, vals, conn, _ = hip.mapping_matrix(gd, gi, n_ref, kern, want_conn=True)
        assert_csr_close(indptr, cols, conn, golden_csr(g, f"conn_{kern}"), rtol=2e-6 if f32_dist else 1e-12)
        if f"map_{kern}_data" in g:
            assert_csr_close(indptr, cols, vals, golden_csr(g, f"map_{kern}"), rtol=1e-5)
    for kern in KERNELS:
        if f"map_{kern}_data" not in g:
            continue
        gm = golden_csr(g, f"map_{kern}")
        gm.sort_indices()
        ip, cc, vv = gm.indptr.astype(np.int64), gm.indices.astype(np.int32), gm.data
        code, conf = hip.label_vote(ip, cc, vv, g["label_codes"], len(g["categories"]))
        cat_code = np.array([int(s[1:]) for s in g["categories"]])
        assert np.array_equal(cat_code[code], g[f"pred_code_{kern}"])
        assert np.array_equal(conf, g[f"conf_{kern}"])
        num = hip.spmm_dense(ip, cc, vv, g["num_values"])
        assert num.dtype == np.float64
        np.testing.assert_array_equal(num, g[f"num_pred_{kern}"])
        om = hip.spmm_dense(ip, cc, vv, g["obsm_in"])
        assert om.dtype == np.float32
        np.testing.assert_array_equal(om, g[f"obsm_pred_{kern}"])
        ld = hip.spmm_dense(ip, cc, vv, g["X_dense"])
        np.testing.assert_array_equal(ld, g[f"layer_dense_{kern}"])
        Xs = golden_csr(g, "X_sparse")
        oi, oc, ov = hip.spmm_csr(ip, cc, vv, Xs.indptr, Xs.indices, Xs.data, Xs.shape[1])
        gs = golden_csr(g, f"layer_sparse_{kern}")
        ours = sp.csr_matrix((ov, oc, oi), shape=gs.shape)
        assert ours.has_sorted_indices or True
        assert ours.nnz == gs.nnz
        np.testing.assert_array_equal(ours.toarray(), gs.toarray())


def test_small_and_padded_graphs(hip):
    g = load_golden("small_data")
    dist, idx = hip.knn(g["xrep"], g["yrep"], 3)
    assert_knn_parity(dist, idx, g["yx_dist"], g["yx_idx"], g["xrep"], g["yrep"])
    for kern in KERNELS:
        indptr, cols, _, conn, _ = hip.mapping_matrix(g["yx_dist"], g["yx_idx"], 5, kern, want_conn=True)
        assert_csr_close(indptr, cols, conn, golden_csr(g, f"conn_{kern}"), rtol=1e-14)
    g = load_golden("sample_graph")
    for kern in KERNELS:
        indptr, cols, vals, conn, _ = hip.mapping_matrix(g["distances"], g["indices"], 3, kern, want_conn=True)
        assert_csr_close(indptr, cols, conn, golden_csr(g, f"conn_{kern}"), rtol=1e-14)
        assert_csr_close(indptr, cols, vals, golden_csr(g, f"map_{kern}"), rtol=2e-7)
    g = load_golden("padded_graph")
    for tag in ("none", "true", "false"):
        gd, gi = g[f"dist_{tag}"], g[f"idx_{tag}"]
        for kern in KERNELS:
            indptr, cols, vals, conn, _ = hip.mapping_matrix(gd, gi, gi.shape[0], kern, want_conn=True)
            assert_csr_close(indptr, cols, conn, golden_csr(g, f"conn_{kern}_{tag}"), rtol=1e-13)
            assert_csr_close(indptr, cols, vals, golden_csr(g, f"map_{kern}_{tag}"), rtol=2e-7)


@pytest.mark.parametrize("n_ref,nq,d,k,dtype", [
    (20000, 3000, 50, 30, np.float32),   # several reference tiles, one split
    (70000, 300, 50, 15, np.float32),    # few query blocks -> reference splits + multi-split re-rank
    (5000, 1000, 128, 30, np.float32),   # config-4 embedding width
    (4000, 777, 37, 10, np.float64),     # float64 input, ragged sizes
    (3000, 500, 17, 90, np.float32),     # large k (fewer waves per workgroup)
    (129, 65, 20, 129, np.float32),      # k == n_ref
])
def test_knn_matches_oracle(n_ref, nq, d, k, dtype, hip, oracle):
    rng = np.random.default_rng(n_ref + nq)
    ref = rng.standard_normal((n_ref, d)).astype(dtype)
    qry = rng.standard_normal((nq, d)).astype(dtype)
    qry[: min(10, nq)] = ref[: min(10, nq)]  # exact self matches
    dist, idx = hip.knn(ref, qry, k)
    od, oi = oracle.knn_euclidean(ref, qry, k)
    assert np.array_equal(idx, oi), "neighbour indices must be bit-exact vs the float64 CPU oracle"
    assert np.array_equal(dist, od), "float64 distances follow the same arithmetic: bit-exact"
    st = hip.knn_stats()
    assert st["fallback_queries"] <= max(2, nq // 100), st


@pytest.mark.parametrize("n,nq,d,k,dtype,scale", [
    (30000, 4000, 2, 90, np.float32, 2000.0),   # spatial coordinates (BASELINE config 5 shape), self-style
    (20000, 2500, 2, 30, np.float64, 2000.0),   # float64 coordinates
    (8000, 1500, 3, 15, np.float32, 50.0),
    (6000, 1000, 10, 15, np.float32, 1.0),      # still sklearn's kd_tree regime (d <= 15)
    (5000, 700, 15, 40, np.float64, 1.0),
])
def test_lowd_direct_path_matches_oracle(n, nq, d, k, dtype, scale, hip, oracle):
    rng = np.random.default_rng(n + d)
    ref = (rng.uniform(0, scale, (n, d)) if d <= 3 else rng.standard_normal((n, d)) * scale).astype(dtype)
    qry = ref[rng.choice(n, nq, replace=False)].copy() if d <= 3 else rng.standard_normal((nq, d)).astype(dtype)
    dist, idx = hip.knn(ref, qry, k)
    od, oi = oracle.knn_euclidean(ref, qry, k)  # direct (kd_tree) form for d <= 15
    assert np.array_equal(idx, oi)
    assert np.array_equal(dist, od)
    st = hip.knn_stats()
    assert st["lowd_path"] == 3, st
    assert st["fallback_queries"] <= max(2, nq // 200), st


@pytest.mark.parametrize("n_ref,nq,d,k,dtype", [
    (1, 1, 20, 1, np.float32),        # smallest possible problem
    (7, 3, 1, 7, np.float64),         # d = 1, k = n_ref
    (300, 1, 50, 30, np.float32),     # single query
    (2000, 600, 61, 20, np.float32),  # widest embedding of the split-fp16 kernel (61 + 3 norm slots = 64)
    (2000, 600, 62, 20, np.float32),  # first width served by the f32-input MFMA kernel
    (1500, 300, 200, 10, np.float32), # wide embedding (f32-input MFMA kernel)
    (9000, 1300, 128, 30, np.float32),      # widest embedding of the K=128 split-fp16 kernel (config 4 width)
    (40000, 700, 100, 15, np.float64),      # K=128 kernel, float64 input, reference splits
    (3000, 300, 129, 10, np.float32),       # first width beyond it (f32-input MFMA kernel)
    (4000, 500, 50, 100, np.float32), # k = 100: K' = 125, 512-entry merges
    (3000, 200, 30, 300, np.float32), # k beyond the in-register merge: f32-input kernel with LDS candidate buffers
    (2500, 400, 16, 5, np.float32),   # smallest d on the MFMA path (d > 15 -> GEMM-form distances)
])
def test_knn_edge_shapes_match_oracle(n_ref, nq, d, k, dtype, hip, oracle):
    rng = np.random.default_rng(1000 * d + k)
    ref = rng.standard_normal((n_ref, d)).astype(dtype)
    qry = rng.standard_normal((nq, d)).astype(dtype)
    dist, idx = hip.knn(ref, qry, k)
    od, oi = oracle.knn_euclidean(ref, qry, k)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_knn_empty_query_and_offsets(hip):
    ref = np.random.default_rng(0).standard_normal((100, 20)).astype(np.float32)
    dist, idx = hip.knn(ref, np.zeros((0, 20), np.float32), 5)
    assert dist.shape == (0, 5) and idx.shape == (0, 5)
    indptr, cols, vals, conn, _ = hip.mapping_matrix(np.zeros((4, 3)), np.array([[0, 1, -1]] * 4), 5, "equal", want_conn=True)
    assert indptr.tolist() == [0, 2, 4, 6, 8] and vals.tolist() == [0.5] * 8


def test_mixed_scale_data_stays_exact(hip, oracle):
    """Coordinates spanning 12 orders of magnitude and a large common offset: the split-fp16 filter loses precision,
    the certificate notices and the float64 stages keep the result exact."""
    rng = np.random.default_rng(77)
    ref = rng.standard_normal((5000, 40)) * np.logspace(-6, 6, 40)[None, :]
    ref[:, 0] += 1e7
    qry = ref[rng.choice(5000, 300, replace=False)] + rng.standard_normal((300, 40)) * 1e-3
    for dt in (np.float64, np.float32):
        r, q = ref.astype(dt), qry.astype(dt)
        dist, idx = hip.knn(r, q, 10)
        od, oi = oracle.knn_euclidean(r, q, 10)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_knn_forced_fallback_and_duplicates(hip, oracle):
    from cellmapper_amd import _native
    import ctypes as C

    rng = np.random.default_rng(5)
    ref = rng.standard_normal((3000, 50)).astype(np.float32)
    ref[100:180] = ref[100]  # 80 identical rows: more ties than K' candidates -> certification must fail
    qry = rng.standard_normal((200, 50)).astype(np.float32)
    qry[0] = ref[100]
    dist, idx = hip.knn(ref, qry, 30)
    od, oi = oracle.knn_euclidean(ref, qry, 30)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert hip.knn_stats()["fallback_queries"] >= 1


def test_mass_duplicates_within_the_kth_distance(hip, oracle):
    """3 000 identical reference rows: every query that sits on them has 3 000 rows tied at its k-th distance -- beyond every
    candidate list, so the exact fallback collects them all (capacity 4 096 per query) and the lowest indices win, like the
    oracle's (value, index) order.  5 000 of them overflow the collection: those queries take the selection kernel (k
    passes over the reference, no collection at all) -- the reference (sklearn) accepts such inputs, so must this."""
    rng = np.random.default_rng(99)
    d, k = 24, 10
    ref = rng.standard_normal((20000, d)).astype(np.float32)
    dup = rng.choice(20000, 3000, replace=False)
    ref[dup] = ref[dup[0]]
    qry = rng.standard_normal((300, d)).astype(np.float32)
    qry[:5] = ref[dup[0]]
    dist, idx = hip.knn(ref, qry, k)
    od, oi = oracle.knn_euclidean(ref, qry, k)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi)
    assert np.array_equal(idx[0], np.sort(dup)[:k]) and np.all(dist[:5] == 0)
    dup2 = rng.choice(20000, 5000, replace=False)
    ref[dup2] = ref[dup2[0]]
    qry[:5] = ref[dup2[0]]
    qry[200] = ref[dup2[0]] + np.float32(1e-3)   # just off the pile: 5 000 rows tied at its FIRST distance
    dist, idx = hip.knn(ref, qry, k)
    od, oi = oracle.knn_euclidean(ref, qry, k)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi)
    assert np.array_equal(idx[0], np.sort(dup2)[:k]) and np.all(dist[:5] == 0)
    # the same through the forced exact scan (sliced scan + rank kernels for the first 1 024 queries, then one workgroup each)
    import torch
    from cellmapper_amd.device import DevicePipeline
    pipe = DevicePipeline(0)
    with torch.cuda.stream(pipe.stream):
        dd, di = pipe.knn(torch.from_numpy(ref).cuda(), torch.from_numpy(np.tile(qry, (5, 1))).cuda(), k, force_fallback=True)
        torch.cuda.synchronize()
    assert np.array_equal(dd.cpu().numpy(), np.tile(od, (5, 1))) and np.array_equal(di.cpu().numpy(), np.tile(oi, (5, 1)))


def test_exact_fallback_sliced_and_per_query_paths(oracle):
    """CM_KNN_FORCE_FALLBACK sends every query through the fp64 exact scan: the first 1024 flagged queries take the
    sliced scan + rank kernels, the rest one workgroup per query -- both must reproduce the oracle bit for bit."""
    import torch
    from cellmapper_amd.device import DevicePipeline

    rng = np.random.default_rng(17)
    ref = rng.standard_normal((5000, 24)).astype(np.float32)
    ref[40:60] = ref[40]
    qry = rng.standard_normal((1300, 24)).astype(np.float32)
    qry[7] = ref[40]
    pipe = DevicePipeline(0)
    with torch.cuda.stream(pipe.stream):
        dist, idx = pipe.knn(torch.from_numpy(ref).cuda(), torch.from_numpy(qry).cuda(), 12, force_fallback=True)
        torch.cuda.synchronize()
    assert pipe.h.knn_stats()["fallback_queries"] == 1300
    od, oi = oracle.knn_euclidean(ref, qry, 12)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(dist.cpu().numpy(), od)


@pytest.mark.parametrize("sigma", [1.0, 0.2, 0.02])
def test_staged_search_on_clustered_embeddings(hip, oracle, sigma):
    """Tight, off-centre clusters are the hard case of the |y|^2 - 2q.y filter (cancellation): the staged search must
    climb from the split-fp16 levels to the direct-difference level / exact scan and stay bit-exact.  nq >= 32768 so
    that the pilot picks the start level."""
    rng = np.random.default_rng(23)
    d, k, n_ref, nq = 50, 15, 12000, 32768
    cen = rng.standard_normal((12, d)) * 5.0
    ref = (cen[rng.integers(0, 12, n_ref)] + sigma * rng.standard_normal((n_ref, d))).astype(np.float32)
    qry = (cen[rng.integers(0, 12, nq)] + sigma * rng.standard_normal((nq, d))).astype(np.float32)
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    od, oi = oracle.knn_euclidean(ref, qry, k)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od), st
    if sigma <= 0.02:
        assert st["filter_products"] >= 3, st   # the hi*hi level cannot certify these: the pilot must skip it


def test_staged_search_float64_and_cosine_clusters(hip, oracle):
    """Same hard case without the pilot (nq < 32768: the ladder starts at level 1 and escalates subsets): float64
    Euclidean input (bit-exact) and float32 cosine (sklearn's float32 GEMM semantics: tie-aware comparison)."""
    rng = np.random.default_rng(29)
    d, k, n_ref, nq = 40, 12, 9000, 3000
    cen = rng.standard_normal((8, d)) * 4.0
    ref = cen[rng.integers(0, 8, n_ref)] + 0.05 * rng.standard_normal((n_ref, d))
    qry = cen[rng.integers(0, 8, nq)] + 0.05 * rng.standard_normal((nq, d))
    dist, idx = hip.knn(ref, qry, k)
    od, oi = oracle.knn_euclidean(ref, qry, k)
    assert dist.dtype == np.float64 and np.array_equal(idx, oi) and np.array_equal(dist, od), hip.knn_stats()
    # cosine: clusters wide enough that the float32 GEMM of the reference semantics still resolves the neighbours
    ref = cen[rng.integers(0, 8, n_ref)] + 1.0 * rng.standard_normal((n_ref, d))
    qry = cen[rng.integers(0, 8, nq)] + 1.0 * rng.standard_normal((nq, d))
    r32, q32 = ref.astype(np.float32), qry.astype(np.float32)
    dist, idx = hip.knn(r32, q32, k, metric="cosine")
    od, oi = oracle.knn_cosine(r32, q32, k)
    assert_knn_parity(dist, idx, od, oi, r32, q32, "cosine")
    # and against an exact float64 evaluation of the same definition (no float32 GEMM noise): identical neighbour sets
    rn = ref.astype(np.float32).astype(np.float64); qn = qry.astype(np.float32).astype(np.float64)
    rn /= np.linalg.norm(rn, axis=1, keepdims=True); qn /= np.linalg.norm(qn, axis=1, keepdims=True)
    ex = np.argsort(1.0 - qn[:200] @ rn.T, axis=1, kind="stable")[:, :k]
    assert all(set(ex[i]) == set(idx[i]) for i in range(200))


def test_staged_search_stale_level_hint_is_harmless(hip, oracle):
    """The handle remembers that the hi*hi level certified (almost) everything on the previous call and then skips the
    pilot.  A following call on data where that level fails must still be exact (it only wastes the first pass), and a
    third call must have dropped the hint."""
    rng = np.random.default_rng(31)
    d, k = 50, 10
    easy_r = rng.standard_normal((20000, d)).astype(np.float32)
    easy_q = rng.standard_normal((32768, d)).astype(np.float32)
    hip.knn(easy_r, easy_q, k)                      # sets the hint (level 1 works on standard-normal rows)
    cen = rng.standard_normal((10, d)) * 6.0
    hard_r = (cen[rng.integers(0, 10, 15000)] + 0.05 * rng.standard_normal((15000, d))).astype(np.float32)
    hard_q = (cen[rng.integers(0, 10, 32768)] + 0.05 * rng.standard_normal((32768, d))).astype(np.float32)
    od, oi = oracle.knn_euclidean(hard_r, hard_q, k)
    for _ in range(2):                               # first call: stale hint; second: pilot again
        dist, idx = hip.knn(hard_r, hard_q, k)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od), hip.knn_stats()
    assert hip.knn_stats()["filter_products"] >= 3


@pytest.mark.parametrize("d,k,dtype", [(2, 15, np.float32), (2, 90, np.float32), (3, 20, np.float32), (1, 8, np.float32),
                                       (2, 30, np.float64)])
def test_grid_search_matches_oracle(hip, oracle, d, k, dtype):
    """Spatial coordinates (d <= 3, n_ref >= 4096) take the uniform-grid candidate search: uniform points, a dense blob
    (cells beyond the per-query budget -> brute-force subset), exact duplicates (ties beyond K'), queries outside the
    reference's bounding box, float64 coordinates of magnitude 1e4."""
    rng = np.random.default_rng(100 + d + k)
    n_ref, nq = 30000, 6000
    ref = rng.random((n_ref, d)) * 2000.0
    ref[:4000] = 1000.0 + rng.standard_normal((4000, d)) * 0.5      # dense blob: > 2048 points in a few cells
    ref[5000:5200] = ref[5000]                                      # 200 exact duplicates
    qry = rng.random((nq, d)) * 2400.0 - 200.0                      # some queries outside the bounding box
    qry[:500] = 1000.0 + rng.standard_normal((500, d)) * 0.5
    qry[500] = ref[5000]
    if dtype == np.float64:
        ref += 1.0e4
        qry += 1.0e4
    ref, qry = ref.astype(dtype), qry.astype(dtype)
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    od, oi = oracle.knn_euclidean(ref, qry, k)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od), st
    assert st["lowd_path"] == 3


def test_grid_search_self_mapping_and_brute_agree(hip, oracle, monkeypatch):
    """Self-mapping on a jittered lattice (many near-ties), grid vs the brute-force direct kernel (CELLMAP_LOWD=brute)."""
    rng = np.random.default_rng(77)
    g = np.stack(np.meshgrid(np.arange(150), np.arange(150)), -1).reshape(-1, 2).astype(np.float32)
    pts = (g * 10.0 + rng.standard_normal(g.shape).astype(np.float32) * 0.01).astype(np.float32)
    d1, i1 = hip.knn(pts, pts, 12)
    monkeypatch.setenv("CELLMAP_LOWD", "brute")
    d2, i2 = hip.knn(pts, pts, 12)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    od, oi = oracle.knn_euclidean(pts, pts[:3000], 12)
    assert np.array_equal(i1[:3000], oi) and np.array_equal(d1[:3000], od)


@pytest.mark.parametrize("d", [16, 20, 29, 30, 45, 61, 62, 100, 125, 128])
def test_every_filter_level_is_exact_for_every_kernel_width(hip, oracle, monkeypatch, d):
    """Pin each filter level in turn (CELLMAP_FILTER_PRODUCTS: no escalation, uncertified queries go to the exact scan) for
    feature widths on both sides of every kernel boundary: K = 32 (d <= 29), K = 64 (d <= 61), K = 128 (d <= 128; hi*hi
    on knn_cand_f16<8,1>, three products on knn_cand_f16w), and the padded direct kernel."""
    rng = np.random.default_rng(d)
    ref = rng.standard_normal((9000, d)).astype(np.float32)
    qry = rng.standard_normal((1500, d)).astype(np.float32)
    od, oi = oracle.knn_euclidean(ref, qry, 20)
    for level in ("1", "2", "3", "4"):
        monkeypatch.setenv("CELLMAP_FILTER_PRODUCTS", level)
        monkeypatch.setenv("CELLMAP_NO_HINT", "1")
        dist, idx = hip.knn(ref, qry, 20)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od), (d, level, hip.knn_stats())


@pytest.mark.parametrize("sigma,metric", [(1.0, "euclidean"), (0.3, "euclidean"), (1.0, "cosine")])
def test_bucket_centred_level_is_exact(hip, oracle, monkeypatch, sigma, metric):
    """The bucket-centred hi*hi level (CELLMAP_FILTER_PRODUCTS=5; chosen by the pilot for >= 131k clustered queries):
    queries packed by bucket, q' - c in the B operand, float64-computed side values as accumulator init.  Pinned here on a
    smaller clustered problem; results must equal the oracle and the plain path bit for bit."""
    rng = np.random.default_rng(53)
    d, k, n_ref, nq = 50, 15, 70000, 8192
    cen = rng.standard_normal((20, d)) * 5.0
    ref = (cen[rng.integers(0, 20, n_ref)] + sigma * rng.standard_normal((n_ref, d))).astype(np.float32)
    qry = (cen[rng.integers(0, 20, nq)] + sigma * rng.standard_normal((nq, d))).astype(np.float32)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    d0, i0 = hip.knn(ref, qry, k, metric=metric)
    monkeypatch.setenv("CELLMAP_FILTER_PRODUCTS", "5")
    d1, i1 = hip.knn(ref, qry, k, metric=metric)
    st = hip.knn_stats()
    assert st["filter_products"] == 5, st
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1), st
    if metric == "euclidean":
        od, oi = oracle.knn_euclidean(ref, qry[:2048], k)
        assert np.array_equal(i1[:2048], oi) and np.array_equal(d1[:2048], od), st
        # the point of centring: far fewer uncertified queries than the plain hi*hi level leaves behind
        monkeypatch.setenv("CELLMAP_FILTER_PRODUCTS", "1")
        hip.knn(ref, qry, k)
        assert st["fallback_queries"] <= hip.knn_stats()["fallback_queries"], (st, hip.knn_stats())


def test_pilot_selects_centred_level_for_large_clustered_sets(hip, oracle, monkeypatch):
    """From 131 072 queries on, a clustered set for which the plain hi*hi level fails makes the pilot start the whole set
    on the bucket-centred level (no forcing here); spot-checked against the oracle."""
    rng = np.random.default_rng(59)
    d, k, n_ref, nq = 50, 15, 70000, 140000
    cen = rng.standard_normal((25, d)) * 5.0
    ref = (cen[rng.integers(0, 25, n_ref)] + 0.5 * rng.standard_normal((n_ref, d))).astype(np.float32)
    qry = (cen[rng.integers(0, 25, nq)] + 0.5 * rng.standard_normal((nq, d))).astype(np.float32)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    assert st["filter_products"] == 5, st
    sel = rng.choice(nq, 1500, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od), st


@pytest.mark.parametrize("metric,sigma,k", [("euclidean", 1.0, 30), ("euclidean", 0.3, 15), ("cosine", 1.0, 20)])
def test_bucket_pruning_on_clustered_sets_is_exact(hip, oracle, monkeypatch, metric, sigma, k):
    """Clustered sets of >= 65 536 queries: the main stage rules out far (query block, reference bucket) pairs with the
    triangle inequality (ivf_* kernels) and visits a fraction of the tiles; results stay identical to the oracle."""
    rng = np.random.default_rng(61)
    d, n_ref, nq = 50, 300000, 262144   # (two 512-query blocks per bucket: the plan leaves well under an eighth of the pairs)
    cen = rng.standard_normal((30, d)) * 5.0
    ref = (cen[rng.integers(0, 30, n_ref)] + sigma * rng.standard_normal((n_ref, d))).astype(np.float32)
    qry = (cen[rng.integers(0, 30, nq)] + sigma * rng.standard_normal((nq, d))).astype(np.float32)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    dist, idx = hip.knn(ref, qry, k, metric=metric)
    st = hip.knn_stats()
    assert 0 < st["tiles_visited_pct"] <= 12, st
    sel = rng.choice(nq, 1200, replace=False)
    if metric == "cosine":
        od, oi = oracle.knn_cosine(ref, qry[sel], k)
    else:
        od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    if metric == "cosine":   # (float32 cosine distances of the oracle carry rounding noise: tie-aware comparison)
        assert_knn_parity(dist[sel], idx[sel], od, oi, ref, qry[sel], "cosine")
    else:
        assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od), st


def test_bucket_pruning_in_subset_stages_float64(hip, oracle, monkeypatch):
    """Very tight, off-centre clusters in float64: the MFMA levels fail for (nearly) every query, the flag lists of
    >= 65 536 queries are regrouped by bucket and the three-product and direct-difference stages prune as well."""
    rng = np.random.default_rng(63)
    d, k, n_ref, nq = 40, 10, 300000, 262144
    cen = rng.standard_normal((30, d)) * 5.0
    ref = cen[rng.integers(0, 30, n_ref)] + 0.02 * rng.standard_normal((n_ref, d))
    qry = cen[rng.integers(0, 30, nq)] + 0.02 * rng.standard_normal((nq, d))
    # a call on easy data first: the handle then starts the clustered set on level 1 without a pilot, so the failures
    # walk down the whole ladder as subset stages
    easy = rng.standard_normal((8192, d))
    hip.knn(easy, easy[:4096], k)
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    assert st["filter_products"] == 1 and st["tiles_visited_pct"] > 0 and st["fallback_queries"] <= 64, st
    sel = rng.choice(nq, 1000, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od), st


@pytest.mark.parametrize("d,k,sigma,dtype,metric", [(50, 30, 1.0, np.float32, "euclidean"), (40, 10, 0.3, np.float64, "euclidean"),
                                                    (50, 15, 0.6, np.float32, "cosine"), (24, 50, 1.0, np.float32, "euclidean")])
def test_chained_pass_b_is_exact(hip, oracle, monkeypatch, d, k, sigma, dtype, metric):
    """Round 5: with four or more query blocks per compute unit the runs of a block's pass B are scanned by ONE work item into
    ONE candidate list (P1Args::wruns; 2 lists per query instead of 14), and the main stage goes further: no pass A at all, the
    plan's radii come from a sample scan and ONE chained pass starts from those seeds.  Both forced here at a test size
    (CELLMAP_DBG=256; +512 = keep pass A): identical to one list per run (CELLMAP_DBG=128) and to the unpruned search on EVERY
    row, identical to the oracle on a sample."""
    from parity import assert_knn_parity

    rng = np.random.default_rng(166 + d)
    n_ref, nq = 260_000, 150_000
    cen = rng.standard_normal((40, d)) * 5.0
    ref = (cen[rng.integers(0, 40, n_ref)] + sigma * rng.standard_normal((n_ref, d))).astype(dtype)
    qry = (cen[rng.integers(0, 40, nq)] + sigma * rng.standard_normal((nq, d))).astype(dtype)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    res = {}
    for tag, dbg in (("chained", "768"), ("single", "256"), ("per_run", "128")):
        monkeypatch.setenv("CELLMAP_DBG", dbg)
        res[tag] = hip.knn(ref, qry, k, metric=metric) + (hip.knn_stats(),)
    monkeypatch.delenv("CELLMAP_DBG")
    monkeypatch.setenv("CELLMAP_NO_IVF", "1")
    res["unpruned"] = hip.knn(ref, qry, k, metric=metric) + (hip.knn_stats(),)
    assert res["chained"][2]["tiles_visited_pct"] > 0 and res["per_run"][2]["tiles_visited_pct"] > 0 and res["unpruned"][2]["tiles_visited_pct"] == 0, \
        [r[2] for r in res.values()]
    assert [res[t][2]["pruning_passes"] for t in ("chained", "single", "per_run", "unpruned")] == [2, 1, 2, 0], [r[2] for r in res.values()]
    for tag in ("single", "per_run", "unpruned"):
        assert np.array_equal(res["chained"][1], res[tag][1]) and np.array_equal(res["chained"][0], res[tag][0]), tag
    sel = rng.choice(nq, 300, replace=False)
    if metric == "cosine":
        od, oi = oracle.knn_cosine(ref, qry[sel], k)
        assert_knn_parity(res["chained"][0][sel], res["chained"][1][sel], od, oi, ref, qry[sel], "cosine")
    else:
        od, oi = oracle.knn_euclidean(ref, qry[sel], k)
        assert np.array_equal(res["chained"][1][sel], oi) and np.array_equal(res["chained"][0][sel], od)


def test_centred_level_retried_with_long_lists(hip, oracle, monkeypatch):
    """Round 5: very tight clusters (sigma 0.1 at a spread of 5).  The bucket-centred level with its k + max(k, 30) candidates
    certifies few queries (their k-th and K'-th neighbours lie closer together than the filter errors); the handle then
    remembers "centred, long lists" and the NEXT call of the same shape runs the level with 80 more candidates, which
    certifies most of them.  Both calls identical on every row, identical to the oracle on a sample."""
    rng = np.random.default_rng(171)
    d, k, n_ref, nq = 50, 30, 300_000, 150_000
    cen = rng.standard_normal((40, d)) * 5.0
    ref = (cen[rng.integers(0, 40, n_ref)] + 0.1 * rng.standard_normal((n_ref, d))).astype(np.float32)
    qry = (cen[rng.integers(0, 40, nq)] + 0.1 * rng.standard_normal((nq, d))).astype(np.float32)
    easy = rng.standard_normal((4096, d + 1)).astype(np.float32)
    hip.knn(easy, easy[:512], k)            # (another shape: the handle forgets what it knew about this one)
    monkeypatch.setenv("CELLMAP_DBG", "256")   # (single-pass pruned main stage at a test size)
    d1, i1 = hip.knn(ref, qry, k)
    st1 = hip.knn_stats()
    d2, i2 = hip.knn(ref, qry, k)
    st2 = hip.knn_stats()
    assert st1["filter_products"] == 5 and st1["candidates"] == 60, st1
    assert st2["filter_products"] == 5 and st2["candidates"] == 140 and st2["pruning_passes"] == 1, st2
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    sel = rng.choice(nq, 400, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(i2[sel], oi) and np.array_equal(d2[sel], od), st2


def test_unpruned_centred_stage_seeds_from_a_centred_sample_scan(hip, oracle, monkeypatch):
    """Round 5 (late): overlapping clusters of unequal size and spread -- the bucket balls overlap, nothing is pruned, and the
    bucket-centred level scans the whole reference.  Its thresholds start from a centred scan of the row sample (side values of
    the sample rows gathered from the full side array).  Identical to the unseeded stage (CELLMAP_DBG=4096) on every row and to
    the oracle on a sample."""
    rng = np.random.default_rng(177)
    d, k, n_ref, nq, ncl = 50, 20, 200_000, 140_000, 30
    cen = rng.standard_normal((ncl, d)) * 5.0
    csig = np.exp(rng.random(ncl) * 2.3 - 1.2)
    aniso = np.exp(rng.standard_normal(d) * 0.5)
    def draw(n, w):
        lab = rng.choice(ncl, n, p=w)
        return (cen[lab] + (0.25 * csig[lab])[:, None] * rng.standard_normal((n, d)) * aniso[None, :]).astype(np.float32)
    ref, qry = draw(n_ref, rng.dirichlet(np.full(ncl, 0.7))), draw(nq, rng.dirichlet(np.full(ncl, 0.7)))
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    monkeypatch.setenv("CELLMAP_NO_IVF", "1")
    d1, i1 = hip.knn(ref, qry, k)
    st1 = hip.knn_stats()
    monkeypatch.setenv("CELLMAP_DBG", "4096")
    d0, i0 = hip.knn(ref, qry, k)
    st0 = hip.knn_stats()
    assert st1["filter_products"] == 5 and st1["tiles_visited_pct"] == 0 and st0["filter_products"] == 5, (st1, st0)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    # ... and the kernel: with pruning known to be off the stage runs on the four-accumulator kernel's centred form (two side blocks
    # per tile, one per 512-query half of its block); CELLMAP_DBG=8192 keeps the two-accumulator kernel
    monkeypatch.setenv("CELLMAP_DBG", "8192")
    d2, i2 = hip.knn(ref, qry, k)
    st2 = hip.knn_stats()
    assert (st1["filter_kernel"], st2["filter_kernel"]) == (2, 1), (st1, st2)
    assert np.array_equal(i2, i1) and np.array_equal(d2, d1)
    sel = rng.choice(nq, 400, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(i1[sel], oi) and np.array_equal(d1[sel], od), st1


def test_single_pass_stage_accepts_large_visited_shares(hip, oracle, monkeypatch):
    """Round 5 (late): overlapping clusters of unequal size and spread.  Pass A + pass B only pay below an eighth of the (block,
    tile) pairs; the single-pass stage pays up to ~79 % (measured break-even) and is accepted up to 70 %.  Forced at a test size
    (CELLMAP_DBG=256): a stage that visits well over an eighth of the pairs, identical to the unpruned search on every row and to
    the oracle on a sample."""
    rng = np.random.default_rng(183)
    d, k, n_ref, nq, ncl = 50, 20, 300_000, 200_000, 40
    cen = rng.standard_normal((ncl, d)) * 5.0
    csig = np.exp(rng.random(ncl) * 2.3 - 1.2)
    aniso = np.exp(rng.standard_normal(d) * 0.5)
    def draw(n, w):
        lab = rng.choice(ncl, n, p=w)
        return (cen[lab] + (0.3 * csig[lab])[:, None] * rng.standard_normal((n, d)) * aniso[None, :]).astype(np.float32)
    ref, qry = draw(n_ref, rng.dirichlet(np.full(ncl, 0.7))), draw(nq, rng.dirichlet(np.full(ncl, 0.7)))
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    monkeypatch.setenv("CELLMAP_DBG", "256")
    d1, i1 = hip.knn(ref, qry, k)
    st1 = hip.knn_stats()
    assert st1["pruning_passes"] == 1 and 13 < st1["tiles_visited_pct"] <= 70, st1
    monkeypatch.delenv("CELLMAP_DBG")
    monkeypatch.setenv("CELLMAP_NO_IVF", "1")
    d0, i0 = hip.knn(ref, qry, k)
    st0 = hip.knn_stats()
    assert st0["tiles_visited_pct"] == 0, st0
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    sel = rng.choice(nq, 400, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(i1[sel], oi) and np.array_equal(d1[sel], od), st1


@pytest.mark.parametrize("prunable", [True, False])
def test_centred_level_from_65536_queries_if_it_can_be_pruned(hip, oracle, monkeypatch, prunable):
    """Round 5 (late): between 16 384 and 131 072 queries the bucket-centred level (whose bucket-aligned blocks prune an order of
    magnitude better than locality-sorted ones) is used when the pruning gate says it will pay, and abandoned for the three-product
    level when it will not (the packing pads such a set by up to a factor of two) -- here: pruning switched off.  Both outcomes
    identical to the oracle."""
    rng = np.random.default_rng(187)
    d, k, n_ref, nq, ncl = 50, 20, 300_000, 90_000, 40
    cen = rng.standard_normal((ncl, d)) * 5.0
    ref = (cen[rng.integers(0, ncl, n_ref)] + 0.5 * rng.standard_normal((n_ref, d))).astype(np.float32)
    qry = (cen[rng.integers(0, ncl, nq)] + 0.5 * rng.standard_normal((nq, d))).astype(np.float32)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    if not prunable:
        monkeypatch.setenv("CELLMAP_NO_IVF", "1")
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    if prunable:
        assert st["filter_products"] == 5 and st["pruning_passes"] == 1 and st["tiles_visited_pct"] <= 10, st
    else:
        assert st["filter_products"] == 3 and st["pruning_passes"] == 0, st      # (abandoned before anything was launched for it)
    sel = rng.choice(nq, 400, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od), st


def test_wide_split_overflow_repeats_the_level(hip, oracle, monkeypatch):
    """Round 5 (late): eight very tight clusters of 37 k rows.  The queries the long-list centred level leaves uncertified start
    their three-product stage from finite seeds and split the reference ~250 ways -- but the seeds' error margins admit more
    than the 1 024 rows the re-rank compacts: those queries repeat the level with ordinary splits (counted in cm_knn_stats)
    instead of moving on to the direct-difference level.  Identical to the search without wide splits, identical to the oracle."""
    rng = np.random.default_rng(181)
    d, k, n_ref, nq = 50, 30, 300_000, 150_000
    cen = rng.standard_normal((8, d)) * 5.0
    ref = (cen[rng.integers(0, 8, n_ref)] + 0.1 * rng.standard_normal((n_ref, d))).astype(np.float32)
    qry = (cen[rng.integers(0, 8, nq)] + 0.1 * rng.standard_normal((nq, d))).astype(np.float32)
    easy = rng.standard_normal((4096, d + 1)).astype(np.float32)
    res = {}
    for tag, dbg in (("wide", "768"), ("plain", "2816")):   # (256 + 512: chained pass A + B at a test size; + 2048: no wide splits)
        hip.knn(easy, easy[:512], k)            # (another shape: the handle forgets what it knew about this one)
        monkeypatch.setenv("CELLMAP_DBG", dbg)
        hip.knn(ref, qry, k)                    # first call: short lists, most queries uncertified -> the handle switches to long lists
        res[tag] = hip.knn(ref, qry, k) + (hip.knn_stats(),)
    assert res["wide"][2]["candidates"] == 140 and res["wide"][2]["wide_split_retries"] >= 1 and res["wide"][2]["subset_splits"] > 16, res["wide"][2]
    assert res["plain"][2]["wide_split_retries"] == 0 and res["plain"][2]["subset_splits"] <= 16, res["plain"][2]
    assert np.array_equal(res["wide"][1], res["plain"][1]) and np.array_equal(res["wide"][0], res["plain"][0])
    sel = rng.choice(nq, 300, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(res["wide"][1][sel], oi) and np.array_equal(res["wide"][0][sel], od)


@pytest.mark.parametrize("sigma", [2.0, 0.3])
def test_small_seeded_subsets_split_the_reference_to_fill_the_chip(hip, oracle, monkeypatch, sigma):
    """Round 5: the few hundred queries a main stage leaves uncertified all carry a finite bound (counted by the re-rank), so
    their stage starts every list from a seed and splits the reference far beyond the 16 ways S x K' <= 1 024 allowed -- the
    re-rank compacts the mostly empty lists.  Identical to the same search without the wide split (CELLMAP_DBG=2048) on every
    row, identical to the oracle on a sample that contains the rows of that stage."""
    rng = np.random.default_rng(173)
    d, k, n_ref, nq = 50, 30, 400_000, 140_000
    cen = rng.standard_normal((40, d)) * 5.0
    ref = (cen[rng.integers(0, 40, n_ref)] + sigma * rng.standard_normal((n_ref, d))).astype(np.float32)
    qry = (cen[rng.integers(0, 40, nq)] + sigma * rng.standard_normal((nq, d))).astype(np.float32)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    monkeypatch.setenv("CELLMAP_SLACK", "6")    # (short lists: a few hundred queries fail the first level's certificate)
    monkeypatch.setenv("CELLMAP_DBG", "2048")
    d0, i0 = hip.knn(ref, qry, k)
    st0 = hip.knn_stats()
    monkeypatch.delenv("CELLMAP_DBG")
    d1, i1 = hip.knn(ref, qry, k)
    st1 = hip.knn_stats()
    assert 0 < st0["subset_splits"] <= 16 and st1["subset_splits"] >= st0["subset_splits"], (st0, st1)
    if sigma == 2.0:   # (a few hundred failures: one or two query blocks; at sigma 0.3 the subset is too large for more splits)
        assert st1["subset_splits"] > 16, (st0, st1)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    sel = rng.choice(nq, 300, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(i1[sel], oi) and np.array_equal(d1[sel], od), st1


@pytest.mark.parametrize("d,sigma,dtype", [(128, 1.0, np.float32), (100, 0.3, np.float32), (72, 2.5, np.float64)])
def test_wide_family_bucket_pruning_is_exact(hip, oracle, monkeypatch, d, sigma, dtype):
    """Round 5: the K = 128 family (62 <= d <= 128) prunes (query block, reference bucket) pairs like the K <= 64 kernels -- the
    hi*hi level on 512-query blocks, the three-product level (knn_cand_f16w, work-list mode) on 256-query blocks, the direct
    level on 256-query blocks.  Clustered sets large enough for the gate: identical to the search without pruning on EVERY row,
    identical to the oracle on a sample; the pruned stage visits a fraction of the (block, tile) pairs."""
    rng = np.random.default_rng(65 + d)
    k, n_ref, nq = 20, 220_000, 140_000
    cen = rng.standard_normal((40, d)) * 5.0
    ref = (cen[rng.integers(0, 40, n_ref)] + sigma * rng.standard_normal((n_ref, d))).astype(dtype)
    qry = (cen[rng.integers(0, 40, nq)] + sigma * rng.standard_normal((nq, d))).astype(dtype)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    assert st["lowd_path"] == 4 and st["fallback_queries"] <= 64, st
    if sigma <= 1.0:
        assert 0 < st["tiles_visited_pct"] <= 13, st
    monkeypatch.setenv("CELLMAP_NO_IVF", "1")
    dist2, idx2 = hip.knn(ref, qry, k)
    st2 = hip.knn_stats()
    assert st2["tiles_visited_pct"] == 0, st2
    assert np.array_equal(idx, idx2) and np.array_equal(dist, dist2), (st, st2)
    # ... and the forms large sets take (chained runs; at the hi*hi level one pass from sample-scan seeds), forced at this size
    monkeypatch.delenv("CELLMAP_NO_IVF")
    monkeypatch.setenv("CELLMAP_DBG", "256")
    dist3, idx3 = hip.knn(ref, qry, k)
    st3 = hip.knn_stats()
    monkeypatch.delenv("CELLMAP_DBG")
    if sigma <= 1.0:
        assert st3["tiles_visited_pct"] > 0 and st3["pruning_passes"] == (1 if st3["filter_products"] == 1 else 2), st3
    assert np.array_equal(idx, idx3) and np.array_equal(dist, dist3), (st, st3)
    sel = rng.choice(nq, 400, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od), st


@pytest.mark.parametrize("d,k,dtype", [(32, 10, np.float32), (50, 30, np.float64), (100, 15, np.float32)])
def test_sample_seeded_thresholds_are_exact(hip, oracle, monkeypatch, d, k, dtype):
    """Unclustered sets over >= 1 024 reference tiles: the main scan starts from thresholds seeded by a
    scan of every 64th reference row and merges only when a log fills; results identical to the oracle, and to the
    unseeded search."""
    rng = np.random.default_rng(64)
    n_ref, nq = 140000, 135000
    ref = rng.standard_normal((n_ref, d)).astype(dtype)
    qry = rng.standard_normal((nq, d)).astype(dtype)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    assert st["fallback_queries"] <= 64, st
    sel = rng.choice(nq, 1000, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od), st
    monkeypatch.setenv("CELLMAP_NO_SEED", "1")
    dist2, idx2 = hip.knn(ref, qry, k)
    assert np.array_equal(idx, idx2) and np.array_equal(dist, dist2)


# the 16x16x32 re-tile (knn_cand_f16r) is part of the experiment build only (round 5: `make ABLATE=1`, run the tests with
# CELLMAP_HIP_LIB=build_ab/lib_ablate.so CELLMAP_ABLATE_BUILD=1); the product library ignores CELLMAP_R16
_R16 = ["1", "0"] if os.environ.get("CELLMAP_ABLATE_BUILD") else ["0"]


@pytest.mark.parametrize("n_ref,nq,d,k,dtype", [
    (140000, 9000, 50, 30, np.float32),     # seeded thresholds (>= 1 024 tiles), reference splits (few query blocks)
    (20000, 5000, 61, 15, np.float64),      # refresh schedule from +inf thresholds, ragged last block
    (150000, 3000, 20, 10, np.float32),     # K = 32 instantiation
    (30000, 40000, 33, 50, np.float32),     # many blocks, wide lists (K' = 75)
])
@pytest.mark.parametrize("r16", _R16)
def test_four_accumulator_filter_is_exact(hip, oracle, monkeypatch, n_ref, nq, d, k, dtype, r16):
    """The 128-queries-per-wave kernels of the hi*hi level -- knn_cand_f16q (32x32x16, accumulator-major chains; K = 32 and
    K = 64) and, in the experiment build, knn_cand_f16r (round 4: v_mfma_f32_16x16x32_f16, four lanes and four logs per
    query, K = 64) -- forced on small problems: identical to the oracle and to the two-accumulator kernel, bit for bit."""
    rng = np.random.default_rng(n_ref + d)
    ref = rng.standard_normal((n_ref, d)).astype(dtype)
    qry = rng.standard_normal((nq, d)).astype(dtype)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    monkeypatch.setenv("CELLMAP_Q4", "2")
    monkeypatch.setenv("CELLMAP_R16", r16)
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    assert st["fallback_queries"] <= 64 and st["filter_products"] == 1, st
    sel = rng.choice(nq, 600, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od), st
    monkeypatch.setenv("CELLMAP_Q4", "0")
    dist2, idx2 = hip.knn(ref, qry, k)
    assert np.array_equal(idx, idx2) and np.array_equal(dist, dist2)


@pytest.mark.skipif(not os.environ.get("CELLMAP_ABLATE_BUILD"), reason="knn_cand_f16r is built by `make ABLATE=1` only")
def test_r16_filter_with_long_logs_and_clusters(hip, oracle, monkeypatch):
    """knn_cand_f16r on inputs that stress its candidate machinery: rows sorted by cluster without the scattered tile order
    (bursts of passing values: logs fill within a tile and merge on the spot), unseeded thresholds, k large enough for
    eight entries per lane in the merges."""
    rng = np.random.default_rng(404)
    d, k, n_ref, nq = 50, 60, 60000, 6000
    cen = rng.standard_normal((12, d)) * 3.0
    lab = np.sort(rng.integers(0, 12, n_ref))
    ref = (cen[lab] + rng.standard_normal((n_ref, d))).astype(np.float32)
    qry = (cen[rng.integers(0, 12, nq)] + rng.standard_normal((nq, d))).astype(np.float32)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1"); monkeypatch.setenv("CELLMAP_Q4", "2"); monkeypatch.setenv("CELLMAP_R16", "1")
    monkeypatch.setenv("CELLMAP_NO_PERM", "1"); monkeypatch.setenv("CELLMAP_NO_SORT", "1"); monkeypatch.setenv("CELLMAP_NO_SEED", "1")
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    od, oi = oracle.knn_euclidean(ref, qry[:800], k)
    assert np.array_equal(idx[:800], oi) and np.array_equal(dist[:800], od), st


def test_bucket_pruning_large_k_on_the_last_level(hip, oracle, monkeypatch):
    """k = 100 on very tight float64 clusters: the direct-difference level is the main stage and prunes; with K' = 125 the
    re-rank can rank 8 lists per query, so the plan must not hand out more -- a query dropped for holding too many
    candidates on the last level would go to the exact scan with a loose bound (found by tests/fuzz/fuzz_paths.py)."""
    rng = np.random.default_rng(65)
    d, k, n_ref, nq = 50, 100, 300000, 262144
    cen = rng.standard_normal((40, d)) * 5.0
    ref = cen[rng.integers(0, 40, n_ref)] + 0.05 * rng.standard_normal((n_ref, d))
    qry = cen[rng.integers(0, 40, nq)] + 0.05 * rng.standard_normal((nq, d))
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    assert st["tiles_visited_pct"] > 0 and st["fallback_queries"] <= 64, st
    sel = rng.choice(nq, 600, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od), st


def test_bucket_pruning_backs_off_on_unclustered_sets(hip, oracle, monkeypatch):
    """Isotropic data: nothing can be ruled out, the plan says so after pass A and the stage runs as a full scan."""
    rng = np.random.default_rng(62)
    d, k, n_ref, nq = 24, 10, 70000, 70000
    ref = rng.standard_normal((n_ref, d)).astype(np.float32)
    qry = rng.standard_normal((nq, d)).astype(np.float32)
    monkeypatch.setenv("CELLMAP_NO_HINT", "1")
    dist, idx = hip.knn(ref, qry, k)
    st = hip.knn_stats()
    assert st["tiles_visited_pct"] == 0, st
    sel = rng.choice(nq, 1000, replace=False)
    od, oi = oracle.knn_euclidean(ref, qry[sel], k)
    assert np.array_equal(idx[sel], oi) and np.array_equal(dist[sel], od), st


def test_mfma_f16_accumulation_bound(hip):
    """The phase-1 certificate assumes that a chain of n v_mfma_f32_32x32x16_f16 instructions returns C + sum(a*b) with
    an absolute error of at most (16 + 2n) * 2^-24 * (|C| + sum|a*b|) (the library uses twice
    that), and (round 4, knn_cand_f16r) a chain of n v_mfma_f32_16x16x32_f16 with at most (32 + 2n) * 2^-24 * the same.  Check the model on this device: random operands over a wide exponent range and designed cases (tiny addends
    next to a large one, cancellation)."""
    import ctypes as C

    def run(a, b, c, chain):
        d = np.empty_like(c)
        hip.call("cm_debug_mfma_f16" if c.shape[1] == 32 else "cm_debug_mfma_f16_k32", a.view(np.uint16).ctypes.data_as(C.c_void_p),
                 b.view(np.uint16).ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
                 d.ctypes.data_as(C.c_void_p), c.shape[0], chain)
        return d

    def check(a, b, c, chain):
        d = run(a, b, c, chain)
        a64, b64, c64 = a.astype(np.float64), b.astype(np.float64), c.astype(np.float64)
        exact = c64 + np.einsum("nsik,nskj->nij", a64, b64)  # products of fp16 are exact in float64
        mag = np.abs(c64) + np.einsum("nsik,nskj->nij", np.abs(a64), np.abs(b64))
        err = np.abs(d.astype(np.float64) - exact)
        per = a.shape[3]   # products per instruction: 16 (32x32x16) or 32 (16x16x32, knn_cand_f16r: the same model with 32)
        ok = err <= (per + 2 * chain) * 2.0 ** -24 * mag + 1e-300
        worst[per] = max(worst.get(per, 0.0), float((err / np.maximum(mag, 1e-300)).max() / 2.0 ** -24))
        assert ok.all(), (per, chain, float((err / np.maximum(mag, 1e-300)).max() / 2.0 ** -24))

    worst = {}
    rng = np.random.default_rng(7)
    for chain in (1, 2, 4, 6, 13):   # 16x16x32: 2 instructions per chain in knn_cand_f16r (K = 64)
        for spread in (0, 3, 8):
            n = 128
            a = (rng.standard_normal((n, chain, 16, 32)) * np.exp2(rng.integers(-spread, spread + 1, (n, chain, 16, 32)))).astype(np.float16)
            b = (rng.standard_normal((n, chain, 32, 16)) * np.exp2(rng.integers(-spread, spread + 1, (n, chain, 32, 16)))).astype(np.float16)
            c = (rng.standard_normal((n, 16, 16)) * np.exp2(rng.integers(-4, 8, (n, 16, 16)))).astype(np.float32)
            check(a, b, c, chain)
    for chain in (1, 4, 8, 12, 25):
        for spread in (0, 3, 8):
            n = 64
            a = (rng.standard_normal((n, chain, 32, 16)) * np.exp2(rng.integers(-spread, spread + 1, (n, chain, 32, 16)))).astype(np.float16)
            b = (rng.standard_normal((n, chain, 16, 32)) * np.exp2(rng.integers(-spread, spread + 1, (n, chain, 16, 32)))).astype(np.float16)
            c = (rng.standard_normal((n, 32, 32)) * np.exp2(rng.integers(-4, 8, (n, 32, 32)))).astype(np.float32)
            check(a, b, c, chain)
    # designed: one addend of 1.0 (as C or as a product) and fifteen / sixteen of 2^-s; -1 + 1 + 2^-s
    S = list(range(10, 30))
    n = len(S)
    a = np.zeros((n, 1, 32, 16), np.float16); b = np.zeros((n, 1, 16, 32), np.float16)
    for t, s in enumerate(S):
        a[t, 0] = np.float16(2.0 ** -(s // 2)); b[t, 0] = np.float16(2.0 ** -(s - s // 2))
    check(a, b, np.ones((n, 32, 32), np.float32), 1)
    a2, b2 = a.copy(), b.copy()
    a2[:, 0, :, 0] = 1.0; b2[:, 0, 0, :] = 1.0
    check(a2, b2, np.zeros((n, 32, 32), np.float32), 1)
    check(a2, b2, -np.ones((n, 32, 32), np.float32), 1)
    # the same designed cases on the 16x16x32 shape: thirty-one / thirty-two addends of 2^-s next to one of 1.0
    a = np.zeros((n, 1, 16, 32), np.float16); b = np.zeros((n, 1, 32, 16), np.float16)
    for t, s in enumerate(S):
        a[t, 0] = np.float16(2.0 ** -(s // 2)); b[t, 0] = np.float16(2.0 ** -(s - s // 2))
    check(a, b, np.ones((n, 16, 16), np.float32), 1)
    a2, b2 = a.copy(), b.copy()
    a2[:, 0, :, 0] = 1.0; b2[:, 0, 0, :] = 1.0
    check(a2, b2, np.zeros((n, 16, 16), np.float32), 1)
    check(a2, b2, -np.ones((n, 16, 16), np.float32), 1)
    print("worst observed error / (2^-24 * sum|terms|) per instruction shape:", worst)


def test_cosine_matches_oracle(hip, oracle):
    rng = np.random.default_rng(11)
    ref = rng.standard_normal((6000, 50)).astype(np.float32)
    qry = rng.standard_normal((500, 50)).astype(np.float32)
    dist, idx = hip.knn(ref, qry, 15, metric="cosine")
    od, oi = oracle.knn_cosine(ref, qry, 15)
    assert dist.dtype == np.float32
    assert_knn_parity(dist, idx, od, oi, ref, qry, "cosine")


@pytest.mark.parametrize("dtype,d,n_ref,nq,k", [(np.float32, 50, 6000, 500, 15), (np.float64, 37, 3000, 300, 30), (np.float32, 200, 2500, 200, 10)])
def test_correlation_matches_oracle(hip, oracle, dtype, d, n_ref, nq, k):
    """sklearn's "correlation" goes to scipy.spatial.distance.cdist: cosine distance of the row-centred data in float64.
    Here: float64 centred copies of both sides, then the cosine search (staged, certified) -- compared with the oracle's
    restatement (pinned on the reference by tests/golden/*correlation*) at the tolerance tests/parity.py states (1e-13)."""
    rng = np.random.default_rng(29)
    ref = (rng.standard_normal((n_ref, d)) + rng.standard_normal((n_ref, 1))).astype(dtype)   # rows with different means
    qry = (rng.standard_normal((nq, d)) + 3.0).astype(dtype)
    dist, idx = hip.knn(ref, qry, k, metric="correlation")
    od, oi = oracle.knn_correlation(ref, qry, k)
    assert dist.dtype == np.float64
    assert_knn_parity(dist, idx, od, oi, ref, qry, "correlation")
    assert len(tie_rows(idx, oi)) == 0


def test_shared_neighbors_match_reference_and_scipy(hip):
    """jaccard / hnoca matrices (cellmapper.py:287-301) vs the golden mapping matrices and vs scipy SpGEMM counts."""
    g = load_golden("synth_full_d37_k10")
    k = int(g["k"])
    for meth in ("jaccard", "hnoca"):
        ip, cc, vv = hip.shared_neighbors(g["yx_idx"], g["xx_idx"], g["yy_idx"], g["xy_idx"], meth)
        norm = hip.csr_normalize(ip, vv)
        assert_csr_close(ip, cc, norm, golden_csr(g, f"map_{meth}"), rtol=1e-6)
    rng = np.random.default_rng(4)
    nq, nr, k = 700, 900, 25
    yx, yy = rng.integers(0, nr, (nq, k)), rng.integers(0, nq, (nq, k))
    xx, xy = rng.integers(0, nr, (nr, k)), rng.integers(0, nq, (nr, k))
    xx[:, 0] = 7  # a hub: reference cell 7 is everybody's neighbour -> one very long transposed list
    yx[5, 3] = -1  # padded entries are skipped

    # upstream builds adjacencies by summing duplicate (row, col) pairs, so a repeated neighbour weighs twice there too
    def adj_sum(ix, ncols):
        r = np.repeat(np.arange(ix.shape[0]), ix.shape[1])
        m = ix.ravel() >= 0
        return sp.csr_matrix((np.ones(m.sum()), (r[m], ix.ravel()[m])), shape=(ix.shape[0], ncols))
    ref = (adj_sum(yx, nr) @ adj_sum(xx, nr).T + adj_sum(yy, nq) @ adj_sum(xy, nq).T).tocsr()
    ref.sort_indices()
    ip, cc, vv = hip.shared_neighbors(yx, xx, yy, xy, "counts")
    ours = sp.csr_matrix((vv, cc, ip), shape=(nq, nr))
    assert np.array_equal(ours.indptr, ref.indptr) and np.array_equal(ours.indices, ref.indices)
    np.testing.assert_array_equal(ours.data, ref.data)


def test_spmm_dense_column_slabs(hip, oracle, monkeypatch):
    """Payloads wider than the device budget are processed in column slabs (hipMemcpy2D); same bits as one pass."""
    rng = np.random.default_rng(8)
    n_ref, nq, k, G = 3000, 700, 12, 1000
    cols = np.sort(rng.integers(0, n_ref, (nq, k)), axis=1).astype(np.int32).ravel()
    vals = rng.random(nq * k, dtype=np.float32)
    indptr = np.arange(nq + 1, dtype=np.int64) * k
    for dt in (np.float32, np.float64):
        X = rng.random((n_ref, G)).astype(dt)
        want = oracle.spmm_dense(indptr, cols, vals, X)
        np.testing.assert_array_equal(hip.spmm_dense(indptr, cols, vals, X), want)
        monkeypatch.setenv("CELLMAP_SPMM_BUDGET_MB", "4")  # 4 MB -> several 64-column-aligned slabs
        np.testing.assert_array_equal(hip.spmm_dense(indptr, cols, vals, X), want)
        monkeypatch.delenv("CELLMAP_SPMM_BUDGET_MB")


def test_error_conventions(hip):
    x = np.zeros((5, 20), np.float32)
    with pytest.raises(ValueError, match="n_neighbors <= n_samples_fit"):
        hip.knn(x, x, 6)
    with pytest.raises(ValueError, match="No finite distances"):
        hip.mapping_matrix(np.full((2, 2), np.inf), np.full((2, 2), -1), 2)
    with pytest.raises(ValueError, match="Unknown kernel"):
        hip.mapping_matrix(np.zeros((2, 2)), np.zeros((2, 2), np.int64), 2, kernel="nope")


def test_pipeline_matches_oracle_end_to_end(hip, oracle):
    """k-NN -> gaussian -> normalise -> label / obsm / layer transfer, compared with the oracle (1e-5 rel)."""
    rng = np.random.default_rng(21)
    n_ref, nq, d, k, G = 30000, 4000, 50, 30, 512
    ref = rng.standard_normal((n_ref, d)).astype(np.float32)
    qry = rng.standard_normal((nq, d)).astype(np.float32)
    codes = rng.integers(0, 20, n_ref).astype(np.int32)
    obsm = rng.standard_normal((n_ref, 2)).astype(np.float32)
    X = rng.random((n_ref, G), dtype=np.float32)
    o = oracle.run_path(ref, qry, k, labels=codes, n_cat=20, obsm=obsm, layer=X)
    dist, idx = hip.knn(ref, qry, k)
    assert np.array_equal(idx, o["idx"])
    indptr, cols, vals, _, sigma = hip.mapping_matrix(dist, idx, n_ref, "gaussian")
    np.testing.assert_allclose(sigma, o["param"], rtol=1e-12)
    assert np.array_equal(cols, o["cols"])
    np.testing.assert_allclose(vals, o["vals"], rtol=1e-5)
    code, conf = hip.label_vote(indptr, cols, vals, codes, 20)
    np.testing.assert_allclose(conf, o["conf"], rtol=1e-5)
    differ = code != o["pred_code"]
    assert differ.mean() < 1e-3  # only where the top-2 class weights are within rounding of each other
    np.testing.assert_allclose(hip.spmm_dense(indptr, cols, vals, obsm), o["obsm_pred"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(hip.spmm_dense(indptr, cols, vals, X), o["layer_pred"], rtol=1e-5, atol=1e-6)


def test_full_size_properties_config2(oracle, monkeypatch):
    """BASELINE configs[1] at full size (1.5 M x 1.5 M cells, d=50, k=30), through the device tier of the C ABI, checked by
    size-independent properties: rows sorted, indices valid and distinct, self-mapping returns every cell as its own
    nearest neighbour at distance 0, two different candidate-generation paths (seeded / unseeded thresholds) give
    bit-identical outputs, and a random sample of rows equals the oracle."""
    import torch
    from cellmapper_amd.device import DevicePipeline

    pipe = DevicePipeline(0)
    torch.cuda.set_stream(pipe.stream)
    g = torch.Generator(device=pipe.device)
    g.manual_seed(11)
    n, d, k = 1_500_000, 50, 30
    ref = torch.randn((n, d), generator=g, device=pipe.device)
    qry = torch.randn((n, d), generator=g, device=pipe.device)
    dist, idx = pipe.knn(ref, qry, k)
    torch.cuda.synchronize()
    st = pipe.h.knn_stats()
    assert st["fallback_queries"] <= 64, st
    assert bool((dist[:, 1:] >= dist[:, :-1]).all()) and bool((dist >= 0).all())
    assert bool(((idx >= 0) & (idx < n)).all())
    srt = torch.sort(idx, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())            # k distinct neighbours per row
    monkeypatch.setenv("CELLMAP_NO_SEED", "1")                 # plain scan from +inf thresholds, refresh schedule
    dist2, idx2 = pipe.knn(ref, qry, k)
    torch.cuda.synchronize()
    assert torch.equal(idx, idx2) and torch.equal(dist, dist2)
    monkeypatch.delenv("CELLMAP_NO_SEED")
    rng = np.random.default_rng(12)
    sel = np.sort(rng.choice(n, 384, replace=False))
    od, oi = oracle.knn_euclidean(ref.cpu().numpy(), qry[torch.as_tensor(sel, device=pipe.device)].cpu().numpy(), k)
    assert np.array_equal(idx[torch.as_tensor(sel, device=pipe.device)].cpu().numpy(), oi)
    assert np.array_equal(dist[torch.as_tensor(sel, device=pipe.device)].cpu().numpy(), od)
    del dist2, idx2
    sd, si = pipe.knn(ref, ref, k)                             # self-mapping
    torch.cuda.synchronize()
    assert bool((si[:, 0] == torch.arange(n, device=pipe.device)).all()) and bool((sd[:, 0] == 0).all())
    # transfer at full size: a constant reference label comes back for every query with confidence 1 (rows of the
    # mapping matrix sum to 1 within float32 rounding); random labels: confidence in (0, 1], codes in range
    const = torch.full((n,), 3, dtype=torch.int32, device=pipe.device)
    out = pipe.map_labels(ref, qry, k, const, 20, kernel="gaussian")
    code, conf = out[0], out[1]
    torch.cuda.synchronize()
    assert bool((code == 3).all()) and bool(((conf - 1).abs() < 1e-5).all())
    labels = torch.randint(0, 20, (n,), generator=g, device=pipe.device, dtype=torch.int32)
    out = pipe.map_labels(ref, qry, k, labels, 20, kernel="gaussian")
    code, conf = out[0], out[1]
    torch.cuda.synchronize()
    assert bool(((code >= 0) & (code < 20)).all()) and bool(((conf > 0) & (conf <= 1 + 1e-5)).all())


@pytest.mark.parametrize("k", [8, 30, 32, 33, 64, 65, 100, 128, 130])
def test_weights_and_vote_row_groupings(hip, oracle, monkeypatch, k):
    """Mapping matrix and label vote over ragged rows (padding, duplicate columns, unsorted columns) for row lengths on
    both sides of the 32-entry boundary (two rows per wave below it, one above): identical structure, float64
    connectivities to 1e-14, normalised float32 weights to 2e-7, vote equal to the oracle's."""
    rng = np.random.default_rng(70 + k)
    nq, n_ref = 5003, 4000
    idx = rng.integers(0, n_ref, (nq, k)).astype(np.int64)
    idx[rng.random((nq, k)) < 0.05] = -1                       # padding in arbitrary positions
    dup = rng.random(nq) < 0.3                                  # rows with a duplicated column
    idx[dup, k - 1] = idx[dup, 0]
    dist = np.sort(rng.random((nq, k)) * 3.0, axis=1)
    dist[idx < 0] = np.inf
    for kern in ("gaussian", "inverse_distance"):
        indptr, cols, vals, conn, _ = hip.mapping_matrix(dist, idx, n_ref, kern, want_conn=True)
        w, _ = oracle.kernel_weights(dist, idx, kern)
        oip, oc, ov32, ov64 = oracle.mapping_matrix(w, dist, idx, n_ref)
        assert np.array_equal(indptr, oip) and np.array_equal(cols, oc)
        np.testing.assert_allclose(conn, ov64, rtol=1e-14)
        np.testing.assert_allclose(vals, ov32, rtol=2e-7)
        # the register-resident kernel (k <= 128) and the LDS kernel run the same arithmetic: identical bits
        monkeypatch.setenv("CELLMAP_WEIGHTS_LDS", "1")
        indptr2, cols2, vals2, conn2, _ = hip.mapping_matrix(dist, idx, n_ref, kern, want_conn=True)
        monkeypatch.delenv("CELLMAP_WEIGHTS_LDS")
        assert np.array_equal(indptr, indptr2) and np.array_equal(cols, cols2) and np.array_equal(vals, vals2) and np.array_equal(conn, conn2)
        codes = rng.integers(0, 11, n_ref).astype(np.int32)
        code, conf = hip.label_vote(indptr, cols, vals, codes, 11)
        ocode, oconf = oracle.label_vote(oip, oc, ov32, codes, 11)
        np.testing.assert_allclose(conf, oconf, rtol=1e-5)
        assert (code != ocode).mean() < 2e-3                    # top-2 class sums within float32 rounding


@pytest.mark.parametrize("metric,dtype,d,n_ref,nq,k", [
    ("manhattan", np.float32, 50, 12000, 700, 30),      # brute force: float32 differences, distance rounded to float32
    ("manhattan", np.float64, 33, 9000, 500, 15),
    ("chebyshev", np.float32, 50, 12000, 700, 30),
    ("chebyshev", np.float64, 64, 6000, 400, 10),       # the widest row with two queries per lane
    ("manhattan", np.float32, 100, 9000, 500, 15),      # one query per lane (65 <= d <= 128)
    ("chebyshev", np.float32, 128, 6000, 300, 30),
    ("manhattan", np.float64, 65, 5000, 300, 10),
    ("manhattan", np.float32, 3, 6000, 6000, 20),       # kd-tree arithmetic: float64 copies of the coordinates
    ("chebyshev", np.float32, 10, 12000, 800, 8),
    ("canberra", np.float32, 50, 12000, 700, 30),       # round 4: sklearn's ratio metrics through the same machinery
    ("canberra", np.float64, 33, 9000, 500, 15),
    ("canberra", np.float32, 100, 9000, 500, 15),
    ("canberra", np.float32, 10, 12000, 800, 8),        # ball-tree arithmetic: float64 copies
    ("braycurtis", np.float32, 50, 12000, 700, 30),
    ("braycurtis", np.float64, 64, 6000, 400, 10),
    ("braycurtis", np.float32, 128, 6000, 300, 30),
    ("braycurtis", np.float32, 3, 6000, 6000, 20),
])
def test_metric_passthrough_is_bit_exact(hip, oracle, metric, dtype, d, n_ref, nq, k):
    """knn.py:343,429 hand `metric` to sklearn: manhattan / chebyshev through the VALU direct filter + float64 re-rank must
    give the oracle's neighbours and distances bit for bit (the oracle is pinned on sklearn by tests/golden/*manhattan*,
    *chebyshev*).  Duplicated reference rows force exact ties (lowest index first) and some queries through the fallback."""
    rng = np.random.default_rng(123 + d)
    ref = (rng.standard_normal((n_ref, d)) * (50.0 if d <= 3 else 1.0)).astype(dtype)
    ref[100:140] = ref[100]
    qry = ref.copy() if nq == n_ref else (rng.standard_normal((nq, d)) * (50.0 if d <= 3 else 1.0)).astype(dtype)
    qry[3] = ref[100]
    if metric in ("canberra", "braycurtis"):   # zeros and sign changes: the 0 / 0 rule and |x| + |y|
        ref[rng.random(ref.shape) < 0.15] = 0.0
        qry[rng.random(qry.shape) < 0.15] = 0.0
        ref[100:140] = ref[100]
        qry[3] = ref[100]
        qry[4] = 0.0                            # an all-zero query: Canberra terms |y| / |y| = 1, Bray-Curtis 1 (0 against a zero row)
    dist, idx = hip.knn(ref, qry, k, metric=metric)
    od, oi = oracle.knn_minkowski_family(ref, qry, k, metric)
    assert dist.dtype == np.float64 and np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert idx[3, 0] == 100 and np.all(dist[3, :min(k, 40)] == 0)
    # aliases as sklearn spells them
    alias = {"manhattan": "cityblock", "chebyshev": "infinity", "canberra": "canberra", "braycurtis": "braycurtis"}[metric]
    d2, i2 = hip.knn(ref[:5000], qry[:200], k, metric=alias)
    d3, i3 = hip.knn(ref[:5000], qry[:200], k, metric=metric)
    assert np.array_equal(d2, d3) and np.array_equal(i2, i3)


COUNT_METRICS = ("hamming", "jaccard", "dice", "rogerstanimoto", "russellrao", "sokalmichener", "sokalsneath", "yule")


@pytest.mark.parametrize("metric", COUNT_METRICS)
@pytest.mark.parametrize("dtype,d,n_ref,nq,k", [(np.float32, 40, 5000, 300, 15), (np.float64, 200, 3000, 150, 30),
                                                (np.float32, 10, 4000, 200, 8), (np.float32, 1000, 1500, 64, 10)])
def test_counting_metrics_are_bit_exact(hip, oracle, metric, dtype, d, n_ref, nq, k):
    """Round 4 (late): the counting metrics sklearn takes without parameters (`knn.py:429` hands any name through): hamming on
    the values, the boolean family on `x != 0` -- scipy `cdist` under brute force, sklearn's DistanceMetric on the ball tree,
    one float64 division of the same integer counts either way.  The exact scan gives the oracle's distances AND neighbours
    bit for bit (ties by index on both sides; the oracle is pinned on the reference by tests/golden/bin*), all-false rows
    included (dice / sokalsneath: NaN against another all-false row, ranked last)."""
    rng = np.random.default_rng(900 + d)
    if metric == "hamming":
        ref = rng.integers(0, 3, (n_ref, d)).astype(dtype)
        qry = rng.integers(0, 3, (nq, d)).astype(dtype)
    else:
        dens = 0.3 if d <= 200 else 0.05
        ref = ((rng.random((n_ref, d)) < dens) * rng.uniform(0.5, 3.0, (n_ref, d))).astype(dtype)
        qry = ((rng.random((nq, d)) < dens) * rng.uniform(0.5, 3.0, (nq, d))).astype(dtype)
        ref[7] = 0; ref[8] = 0; qry[5] = 0            # all-false rows on both sides
    ref[100:120] = ref[100]
    qry[3] = ref[100]
    dist, idx = hip.knn(ref, qry, k, metric=metric)
    st = hip.knn_stats()
    od, oi = oracle.knn_count_family(ref, qry, k, metric)
    assert dist.dtype == np.float64 and st["lowd_path"] == 5
    assert np.array_equal(dist, od, equal_nan=True) and np.array_equal(idx, oi)
    assert idx[3, 0] == 100 or dist[3, 0] == 0


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("bin")])
def test_counting_metric_goldens(hip, name):
    """The reference's own output (tests/golden/bin*, made through `oracle/make_golden.py`): distances bit for bit; neighbours
    up to the order and the choice inside tied distances, which upstream leaves to argpartition / the tree's heap."""
    g = load_golden(name)
    xrep, yrep = rep(g)
    metric = str(g["metric"])
    dist, idx = hip.knn(xrep, yrep, int(g["k"]), metric=metric)
    assert np.array_equal(dist, g["yx_dist"], equal_nan=True)
    assert_knn_parity(dist, idx, g["yx_dist"], g["yx_idx"], xrep, yrep, metric)


@pytest.mark.parametrize("name", [n for n in golden_names() if "haversine" in n or "nan_euclidean" in n])
def test_haversine_and_nan_euclidean_goldens(hip, name):
    """Round 5: the last two parameter-free metric names (`knn.py:343,429` hand any of them to sklearn).  Against the reference's
    own output (tests/golden/latlon*, *nan_euclidean*) at the tolerances tests/parity.py states for them -- libm / numpy / device
    transcendentals differ in the last bit; upstream's nan_euclidean is a float32 cancellation for float32 data -- with the NaN
    pattern (pairs without a shared coordinate) identical."""
    g = load_golden(name)
    xrep, yrep = rep(g)
    metric = str(g["metric"])
    dist, idx = hip.knn(xrep, yrep, int(g["k"]), metric=metric)
    assert dist.dtype == g["yx_dist"].dtype and hip.knn_stats()["lowd_path"] == 5
    assert_knn_parity(dist, idx, g["yx_dist"], g["yx_idx"], xrep, yrep, metric)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_haversine_and_nan_euclidean_match_the_oracle_and_the_class(hip, oracle, dtype):
    """Larger random cases against the oracle's restatements (same tolerances), the error cases (haversine needs two columns;
    NaN only under nan_euclidean; infinities never), the drop-in class and the sharded merge."""
    import pandas as pd

    from cellmapper_amd import CellMapper, MiniAnnData, _native

    rng = np.random.default_rng(5 + (dtype == np.float64))
    ll = np.c_[rng.uniform(-1.5, 1.5, 9000), rng.uniform(-3.1, 3.1, 9000)].astype(dtype)
    lq = np.c_[rng.uniform(-1.5, 1.5, 700), rng.uniform(-3.1, 3.1, 700)].astype(dtype)
    lq[:50] = ll[:50]                                   # exact hits: distance 0
    for k in (10, 5000):                                # ball tree / brute force upstream (k >= n_ref / 2)
        dist, idx = hip.knn(ll, lq[:(700 if k == 10 else 40)], k, metric="haversine")
        od, oi = oracle.knn_haversine(ll, lq[:(700 if k == 10 else 40)], k)
        assert_knn_parity(dist, idx, od, oi, ll, lq, "haversine")
    with pytest.raises(ValueError, match="only valid in 2 dimensions"):
        hip.knn(np.zeros((50, 3), dtype), np.zeros((5, 3), dtype), 3, metric="haversine")
    x = rng.standard_normal((6000, 24)).astype(dtype)
    q = rng.standard_normal((400, 24)).astype(dtype)
    x[rng.random(x.shape) < 0.1] = np.nan
    q[rng.random(q.shape) < 0.1] = np.nan
    q[7] = np.nan                                       # no coordinate at all: every distance NaN
    x[11] = np.nan
    dist, idx = hip.knn(x, q, 12, metric="nan_euclidean")
    od, oi = oracle.knn_nan_euclidean(x, q, 12)
    assert dist.dtype == dtype and np.isnan(dist[7]).all() and not np.isnan(np.delete(dist, 7, 0)).any() and 11 not in np.delete(idx, 7, 0)
    assert_knn_parity(dist, idx, od, oi, x, q, "nan_euclidean")
    with pytest.raises(ValueError, match="NaN"):
        hip.knn(x, q, 12, metric="euclidean")
    xi = x.copy(); xi[3, 3] = np.inf
    with pytest.raises(ValueError, match="infinity"):
        hip.knn(xi, q, 12, metric="nan_euclidean")
    # the drop-in class: NaN representations are accepted under this metric only
    labels = pd.Categorical([f"t{c}" for c in rng.integers(0, 5, x.shape[0])])
    R = MiniAnnData(X=np.zeros((x.shape[0], 1), np.float32), obs=pd.DataFrame({"label": labels}), obsm={"X_emb": x})
    Q = MiniAnnData(X=np.zeros((q.shape[0], 1), np.float32), obsm={"X_emb": q})
    cm = CellMapper(Q, R)
    cm.compute_neighbors(n_neighbors=12, use_rep="X_emb", metric="nan_euclidean", only_yx=True)
    assert cm.knn.yx.distances.dtype == dtype
    assert_knn_parity(cm.knn.yx.distances, cm.knn.yx.indices, od, oi, x, q, "nan_euclidean")
    with pytest.raises(ValueError, match="NaN"):
        CellMapper(Q, R).compute_neighbors(n_neighbors=12, use_rep="X_emb", metric="manhattan", only_yx=True)
    # reference-sharded multi-device graph: NaN values travel through the merge and rank last
    import os
    os.environ["CELLMAP_MULTI_REF_SHARD"] = "1"
    _native.reload_knobs()
    try:
        big = np.tile(x, (3, 1))[:3 * 4096 + 5].copy()
        big += (1e-3 * rng.standard_normal(big.shape)).astype(dtype)
        qq = np.tile(q, (3, 1))[:1100].copy()
        hm = _native.Handle(devices=[0, 0, 0])
        g = _native.DeviceGraph(hm, _native.DeviceRep(hm, big, allow_nan=True), _native.DeviceRep(hm, qq, allow_nan=True), 12, "nan_euclidean")
        assert g.n_devices == 3
        dm, im = g.fetch_knn()
        d1, i1 = hip.knn(big, qq, 12, metric="nan_euclidean")
        assert np.array_equal(dm, d1, equal_nan=True) and np.array_equal(im[~np.isnan(dm)], i1[~np.isnan(d1)])
        g.close(); hm.close()
    finally:
        os.environ.pop("CELLMAP_MULTI_REF_SHARD", None)
        _native.reload_knobs()


@pytest.mark.parametrize("dtype,d,n_ref,nq,k", [(np.float32, 50, 12000, 700, 30), (np.float64, 33, 9000, 500, 15),
                                                (np.float32, 10, 12000, 800, 8), (np.float32, 2, 6000, 6000, 20),
                                                (np.float32, 128, 6000, 300, 30)])
def test_sqeuclidean_is_the_gemm_form_without_the_root(hip, oracle, dtype, d, n_ref, nq, k):
    """sklearn's "sqeuclidean" (knn.py:343,429 pass the name through): brute force whatever d is, float64 ranking values
    returned without the square root -- bit-equal to the oracle (pinned on the reference by tests/golden/*sqeuclidean*), and
    the same neighbours as "euclidean" in its GEMM form."""
    rng = np.random.default_rng(77 + d)
    ref = (rng.standard_normal((n_ref, d)) * (50.0 if d <= 3 else 1.0)).astype(dtype)
    ref[100:110] = ref[100]
    qry = ref.copy() if nq == n_ref else (rng.standard_normal((nq, d)) * (50.0 if d <= 3 else 1.0)).astype(dtype)
    qry[3] = ref[100]
    dist, idx = hip.knn(ref, qry, k, metric="sqeuclidean")
    od, oi = oracle.knn_euclidean(ref, qry, k, squared=True)
    assert dist.dtype == np.float64 and np.array_equal(idx, oi) and np.array_equal(dist, od)
    d2, i2 = hip.knn(ref, qry, k, metric="euclidean", form="gemm")
    assert np.array_equal(i2, idx) and np.array_equal(d2, np.sqrt(dist))


@pytest.mark.parametrize("name", [n for n in golden_names() if "sqeuclidean" in n])
def test_sqeuclidean_matches_reference_golden(name, hip):
    g = load_golden(name)
    xrep, yrep = rep(g)
    dist, idx = hip.knn(xrep, yrep, int(g["k"]), metric="sqeuclidean")
    assert np.array_equal(idx, g["yx_idx"])
    np.testing.assert_allclose(dist, g["yx_dist"], rtol=2e-15, atol=0)   # (BLAS accumulation order upstream: last bit)


@pytest.mark.parametrize("d,n_ref,nq,k,dtype,metric", [
    (260, 6000, 700, 30, np.float32, "euclidean"),      # first width of the K-chunked filter (KP = 384: three chunks)
    (512, 9000, 900, 15, np.float32, "euclidean"),      # foundation-model embedding width
    (512, 5000, 400, 20, np.float64, "euclidean"),
    (1000, 4000, 300, 10, np.float32, "cosine"),
    (2047, 3000, 200, 10, np.float32, "euclidean"),     # the widest supported row
    (700, 40000, 130, 30, np.float32, "euclidean"),     # few query blocks: reference splits + multi-list re-rank
])
def test_wide_rows_chunked_filter_is_exact(hip, oracle, d, n_ref, nq, k, dtype, metric):
    """260 <= d <= 2047 (round 3; `knn.py:429` sets no limit on the width): the f32 MFMA filter with the contraction chunked
    over the feature dimension (`knn_cand_mfma_wide`), float64 re-rank and exact fallback with LDS sized by the row width --
    neighbours and distances bit-equal to the oracle, duplicated reference rows included."""
    rng = np.random.default_rng(d + n_ref)
    ref = rng.standard_normal((n_ref, d)).astype(dtype)
    ref[50:58] = ref[50]
    qry = rng.standard_normal((nq, d)).astype(dtype)
    qry[7] = ref[50]
    dist, idx = hip.knn(ref, qry, k, metric=metric)
    st = hip.knn_stats()
    if metric == "cosine":
        od, oi = oracle.knn_cosine(ref, qry, k)
        assert (np.sort(idx, 1) == np.sort(oi, 1)).mean() > 0.9999 and np.allclose(dist, od, rtol=0, atol=4e-7), st
    else:
        od, oi = oracle.knn_euclidean(ref, qry, k)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od), st
        assert idx[7, 0] == 50 and np.all(dist[7, :8] == 0)
    assert st["lowd_path"] == 1 and st["fallback_queries"] <= max(16, nq // 20), st


@pytest.mark.parametrize("d,n_ref,nq,k,dtype,metric", [
    (2048, 1500, 100, 10, np.float32, "euclidean"),     # first width beyond the chunked filter
    (3000, 1200, 64, 5, np.float64, "euclidean"),
    (5000, 900, 33, 12, np.float32, "sqeuclidean"),
    (2500, 1000, 50, 8, np.float32, "cosine"),
    (2100, 800, 40, 6, np.float32, "correlation"),
    (2300, 30000, 300, 10, np.float32, "euclidean"),     # reference splits, multi-list re-rank
    (9000, 600, 40, 5, np.float64, "euclidean"),         # 72-KB rows
    (130, 3000, 200, 10, np.float32, "manhattan"),      # term-wise metrics beyond the VALU filter's 128 columns
    (300, 2000, 100, 7, np.float64, "chebyshev"),
    (200, 2000, 100, 12, np.float32, "canberra"),
    (150, 2500, 90, 9, np.float64, "braycurtis"),
    (50, 5000, 50, 1500, np.float32, "euclidean"),      # k beyond every filter's candidate buffers
    (200, 3000, 21, 700, np.float64, "euclidean"),
    (20, 4000, 30, 600, np.float32, "manhattan"),
    (64, 4500, 9, 4200, np.float32, "euclidean"),       # the largest single-pass k: one query per workgroup, lists of 6 656
    (16, 12000, 40, 9000, np.float32, "euclidean"),     # round 5: any k -- three passes of 4 096 ranks
    (12, 9000, 9, 8999, np.float64, "manhattan"),
    (7, 300, 70, 300, np.float32, "chebyshev"),         # k == n_ref
])
def test_no_width_limit_and_the_exact_scan(hip, oracle, d, n_ref, nq, k, dtype, metric):
    """Round 4: no width limit, no 128-column limit for the term-wise metrics, k up to 4 200 (`knn.py:429` and sklearn set
    none).  Rows beyond 2 047 columns keep the chunked MFMA filter; the re-rank and the exact fallback read the query row from
    global memory instead of an LDS copy (`GQ`).  Shapes outside every filter family run `knn_exact_scan` -- every pair value
    in float64 by the re-rank's own arithmetic, a running top-k per query in LDS.  All equal the oracle bit for bit,
    duplicated rows included."""
    rng = np.random.default_rng(d * 7 + k)
    ref = rng.standard_normal((n_ref, d)).astype(dtype)
    if metric in ("canberra", "braycurtis"):
        ref[rng.random(ref.shape) < 0.1] = 0
    ref[50:58] = ref[50]
    qry = rng.standard_normal((nq, d)).astype(dtype)
    qry[7] = ref[50]
    dist, idx = hip.knn(ref, qry, k, metric=metric)
    st = hip.knn_stats()
    assert st["lowd_path"] == (1 if d > 2047 else 5) and (d > 2047 or st["fallback_queries"] == 0), st
    if metric == "cosine":
        od, oi = oracle.knn_cosine(ref, qry, k)
        assert (np.sort(idx, 1) == np.sort(oi, 1)).mean() > 0.9999 and np.allclose(dist, od, rtol=0, atol=4e-7)
    elif metric == "correlation":
        od, oi = oracle.knn_correlation(ref, qry, k)
        assert (np.sort(idx, 1) == np.sort(oi, 1)).mean() > 0.9999 and np.allclose(dist, od, rtol=0, atol=1e-13)
    elif metric in ("euclidean", "sqeuclidean"):
        od, oi = oracle.knn_euclidean(ref, qry, k, squared=metric == "sqeuclidean")
        assert np.array_equal(idx, oi) and np.array_equal(dist, od)
        if metric == "euclidean" and k >= 8:
            assert idx[7, 0] == 50 and np.all(dist[7, :8] == 0)
    else:
        od, oi = oracle.knn_minkowski_family(ref, qry, k, metric)
        assert np.array_equal(dist, od)
        ties = dist[:, 1:] == dist[:, :-1]
        assert np.array_equal(idx[:, 1:][~ties], oi[:, 1:][~ties]) or np.array_equal(idx, oi)


def test_exact_scan_passes_split_ties_by_index(hip, oracle):
    """k beyond one pass of the exact scan with thousands of EXACTLY tied rows straddling the pass boundary: a pass ranks only
    the pairs strictly after the previous pass's last (value, index) entry, so a tie group is cut by index, never duplicated or
    dropped."""
    rng = np.random.default_rng(99)
    ref = rng.standard_normal((11000, 12)).astype(np.float32)
    ref[1000:7000] = ref[1000]            # 6 000 identical rows: for every query they tie, across rank 4 096 for most
    qry = rng.standard_normal((9, 12)).astype(np.float32)
    qry[3] = ref[1000]
    dist, idx = hip.knn(ref, qry, 10500)
    assert hip.knn_stats()["lowd_path"] == 5
    od, oi = oracle.knn_euclidean(ref, qry, 10500)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert all(len(set(row)) == len(row) for row in idx)


def test_metric_passthrough_limits_and_names(hip, oracle):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((500, 130)).astype(np.float32)
    with pytest.raises(ValueError, match="not supported"):
        hip.knn(x, x[:10], 5, metric="mahalanobis")
    # magnitudes the ratio metrics' fp32 filter cannot divide by (its reciprocal has no denormal support; a float64 row's
    # float32 image would flush them): the data is screened and takes the exact scan -- same bits as the oracle (round 5;
    # float64 input used to be refused, float32 input could lose a neighbour: round-4 advisor finding)
    for dt, small in ((np.float64, 1e-40), (np.float32, 1e-40), (np.float32, 3e-39), (np.float64, 1e-300)):
        tiny = rng.standard_normal((300, 20)).astype(dt)
        tiny[7, 3] = small
        tiny[9, :] = 0
        tiny[9, 5] = small        # a row whose only non-zero coordinate is subnormal: every Canberra term against it is 0 or 1
        tiny[11, 3] = -small
        for metric in ("canberra", "braycurtis"):
            dist, idx = hip.knn(tiny, tiny[:16], 5, metric=metric)
            assert hip.knn_stats()["lowd_path"] == 5, hip.knn_stats()       # the exact scan ran
            od, oi = oracle.knn_minkowski_family(tiny, tiny[:16], 5, metric)
            assert np.array_equal(dist, od), (dt, small, metric)
            ties = dist[:, 1:] == dist[:, :-1]
            assert np.array_equal(idx[:, 1:][~ties], oi[:, 1:][~ties]) or np.array_equal(idx, oi)
    d0, i0 = hip.knn(x, x[:10], 5, metric="minkowski")     # sklearn's default p = 2
    d1, i1 = hip.knn(x, x[:10], 5, metric="euclidean")
    assert np.array_equal(d0, d1) and np.array_equal(i0, i1)


@pytest.mark.parametrize("d,sigma", [(100, 0.05), (128, 0.02), (70, 0.05)])
def test_wide_rows_reach_the_direct_level_on_tight_clusters(hip, oracle, d, sigma):
    """62 <= d <= 128 (K = 128 kernel family): near-duplicate clusters far from the origin defeat the split-fp16 filters; the
    queries then go through the direct-difference level (one query per lane) instead of the exact scan -- and the result is
    still the oracle's, bit for bit."""
    rng = np.random.default_rng(400 + d)
    cen = rng.standard_normal((12, d)) * 5.0
    ref = (cen[rng.integers(0, 12, 30000)] + sigma * rng.standard_normal((30000, d))).astype(np.float32)
    qry = (cen[rng.integers(0, 12, 3000)] + sigma * rng.standard_normal((3000, d))).astype(np.float32)
    dist, idx = hip.knn(ref, qry, 15)
    st = hip.knn_stats()
    od, oi = oracle.knn_euclidean(ref, qry, 15)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert st["fallback_queries"] < 300, st      # (the exact scan is for the odd query, not for the set)


def test_workspace_covers_small_sets_with_many_neighbours(hip, oracle):
    """Every (width, k) family on tiny query sets: the call's device workspace is sized from the shapes alone, and the fixed
    parts (candidate rows of the staged search at k + slack + 16 per query) dominate here -- a shortfall showed up at d = 128,
    k = 100, one query (tests/fuzz/fuzz_metrics.py).  Results against the oracle."""
    rng = np.random.default_rng(77)
    for d in (2, 16, 50, 61, 62, 100, 128, 200):
        for k in (1, 30, 100, 180):
            for nq, n_ref in ((1, 700), (600, 5000)):
                if k > n_ref or (d > 128 and k > 64):   # (the f32-input kernel of 129 <= d <= 259 keeps its lists in LDS: k <= 64 or so)
                    continue
                ref = rng.standard_normal((n_ref, d)).astype(np.float32)
                qry = rng.standard_normal((nq, d)).astype(np.float32)
                dist, idx = hip.knn(ref, qry, k)
                od, oi = oracle.knn_euclidean(ref, qry, k)
                assert np.array_equal(idx, oi) and np.array_equal(dist, od), (d, k, nq, n_ref)
    # the other metrics run the staged search whatever k is (the sizing does not see the metric)
    for d, k in ((15, 191), (50, 191), (100, 150)):
        ref = rng.standard_normal((6000, d)).astype(np.float32)
        qry = rng.standard_normal((2, d)).astype(np.float32)
        dist, idx = hip.knn(ref, qry, k, metric="manhattan")
        od, oi = oracle.knn_minkowski_family(ref, qry, k, "manhattan")
        assert np.array_equal(idx, oi) and np.array_equal(dist, od), (d, k)
