#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING THE REFERENCE in this container.

TEST INFRASTRUCTURE ONLY.  This script imports quadbio/cellmapper from ``/root/reference``
(read-only; never copied) with two inert stub modules for the packages that are absent
here (``anndata`` -> our MiniAnnData container, ``scanpy`` -> empty module; both are only
touched by out-of-scope fallback code) and drives the reference's own public API:

    Neighbors(xrep, yrep).compute_neighbors(method="sklearn", ...)        knn.py:339-465
    NeighborsResults.knn_graph_connectivities(kernel=...)                  knn.py:134-164
    CellMapper(query, reference).map(use_rep=..., obs_keys=..., ...)       cellmapper.py:426-491

The outputs (plus the inputs and library versions) are stored as small ``.npz`` fixtures
under ``tests/golden/``.  Fixtures are data; the reference itself never ships.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
from __future__ import annotations

import importlib
import importlib.metadata
import os
import sys
import types

import numpy as np
import pandas as pd
import scipy
import scipy.sparse as sp
import sklearn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src/cellmapper"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from cellmapper_amd._anndata import MiniAnnData  # noqa: E402


def load_reference():
    """Import the reference's model modules with inert stubs for anndata / scanpy."""
    pkg = types.ModuleType("cellmapper")
    pkg.__path__ = [REF_SRC]
    pkg.version = importlib.metadata.version  # needed by check.py
    sys.modules["cellmapper"] = pkg
    ad = types.ModuleType("anndata")
    ad.AnnData = MiniAnnData
    sys.modules["anndata"] = ad
    sc = types.ModuleType("scanpy")
    scget = types.ModuleType("scanpy.get")
    scget._check_mask = None
    scget._get_obs_rep = None
    sc.get = scget
    sys.modules["scanpy"] = sc
    sys.modules["scanpy.get"] = scget
    os.environ.setdefault("LOGLEVEL", "WARNING")
    cm = importlib.import_module("cellmapper.model.cellmapper")
    knn = importlib.import_module("cellmapper.model.knn")
    utils = importlib.import_module("cellmapper.utils")
    return cm, knn, utils


VERSIONS = dict(
    v_sklearn=sklearn.__version__, v_scipy=scipy.__version__, v_numpy=np.__version__, v_pandas=pd.__version__
)
KERNELS = ["gaussian", "scarches", "inverse_distance", "equal"]


def csr_parts(prefix, m):
    m = m.tocsr()
    return {f"{prefix}_indptr": m.indptr, f"{prefix}_indices": m.indices, f"{prefix}_data": m.data,
            f"{prefix}_shape": np.asarray(m.shape)}


def label_names(codes):
    return np.array([f"c{c:02d}" for c in codes])


def run_mapping_case(cm_mod, name, xrep, yrep, k, metric="euclidean", seed=0, n_cat=20, G=48, only_yx=True,
                     transfer_kernels=("gaussian",), rep_key="X_emb"):
    """Full CellMapper.map() run through the reference; returns dict of arrays."""
    rng = np.random.default_rng(seed + 1000)
    n_ref, nq = xrep.shape[0], yrep.shape[0]
    label_codes = rng.integers(0, n_cat, n_ref)
    num_values = rng.standard_normal(n_ref)  # float64 numeric obs column
    obsm_in = rng.standard_normal((n_ref, 2)).astype(np.float32)
    X_dense = rng.random((n_ref, G), dtype=np.float32)
    Xs = sp.random(n_ref, G, density=0.1, format="csr", dtype=np.float32, random_state=seed + 7)
    Xs.data = np.exp(rng.standard_normal(Xs.nnz)).astype(np.float32)

    out = dict(xrep=xrep, yrep=yrep, k=np.int64(k), metric=np.array(metric), label_codes=label_codes,
               num_values=num_values, obsm_in=obsm_in, X_dense=X_dense, **csr_parts("X_sparse", Xs), **VERSIONS)

    def make_pair(X):
        ref = MiniAnnData(X=X,
                          obs=pd.DataFrame({"label": pd.Categorical(label_names(label_codes)), "num": num_values},
                                           index=[f"r{i}" for i in range(n_ref)]),
                          obsm={rep_key: xrep, "X_umap": obsm_in})
        qry = MiniAnnData(X=np.zeros((nq, 3), dtype=np.float32),
                          obs=pd.DataFrame(index=[f"q{i}" for i in range(nq)]), obsm={rep_key: yrep})
        return qry, ref

    first = True
    for kern in transfer_kernels:
        qry, ref = make_pair(X_dense)
        cmap = cm_mod.CellMapper(qry, ref)
        cmap.map(obs_keys=["label", "num"], obsm_keys="X_umap", layer_key="X", n_neighbors=k, use_rep=rep_key,
                 knn_method="sklearn", metric=metric, only_yx=only_yx, mapping_method=kern)
        if first:
            out["yx_dist"] = cmap.knn.yx.distances
            out["yx_idx"] = cmap.knn.yx.indices
            for kk in KERNELS:
                out.update(csr_parts(f"conn_{kk}", cmap.knn.yx.knn_graph_connectivities(kernel=kk)))
            if not only_yx:
                for g in ("xx", "yy", "xy"):
                    out[f"{g}_dist"] = getattr(cmap.knn, g).distances
                    out[f"{g}_idx"] = getattr(cmap.knn, g).indices
            first = False
        out.update(csr_parts(f"map_{kern}", cmap.mapping_matrix))
        cats = list(ref.obs["label"].cat.categories)
        pred = qry.obs["label_pred"]
        out[f"pred_code_{kern}"] = np.array([int(s[1:]) for s in pred.astype(str)], dtype=np.int64)
        out[f"conf_{kern}"] = qry.obs["label_conf"].to_numpy()
        out[f"num_pred_{kern}"] = qry.obs["num_pred"].to_numpy()
        out[f"obsm_pred_{kern}"] = np.asarray(qry.obsm["X_umap_pred"])
        out[f"layer_dense_{kern}"] = np.asarray(cmap.query_imputed.X)
        # sparse layer
        qry2, ref2 = make_pair(Xs)
        cmap2 = cm_mod.CellMapper(qry2, ref2)
        cmap2.knn = cmap.knn
        cmap2.only_yx = only_yx
        cmap2.compute_mapping_matrix(method=kern)
        cmap2.map_layers("X")
        res = cmap2.query_imputed.X
        assert sp.issparse(res)
        res = res.tocsr().copy()
        res.sort_indices()
        out.update(csr_parts(f"layer_sparse_{kern}", res))
        out["categories"] = np.array(cats)
    if not only_yx:
        for meth in (("jaccard", "hnoca") if k > 1 else ("jaccard",)):
            qry, ref = make_pair(X_dense)
            cmap3 = cm_mod.CellMapper(qry, ref)
            cmap3.knn = cmap.knn
            cmap3.only_yx = False
            cmap3.compute_mapping_matrix(method=meth)
            mm = cmap3.mapping_matrix.copy()
            mm.sort_indices()
            out.update(csr_parts(f"map_{meth}", mm))
    return out


def presence_case(cm_mod):
    """estimate_presence_score (evaluate.py:426-521) with and without groups, log / percentile variants."""
    rng = np.random.default_rng(11)
    n_ref, nq, d, k = 900, 400, 30, 15
    xrep = rng.standard_normal((n_ref, d)).astype(np.float32)
    yrep = (rng.standard_normal((nq, d)) * 0.7 + 0.3).astype(np.float32)
    groups = np.array(["g%d" % v for v in rng.integers(0, 4, nq)])
    out = dict(xrep=xrep, yrep=yrep, k=np.int64(k), groups=groups, **VERSIONS)
    for tag, kw in (("default", {}), ("log", dict(log=True)), ("nopct", dict(percentile=(0, 100)))):
        ref = MiniAnnData(X=np.zeros((n_ref, 2), np.float32), obs=pd.DataFrame(index=[f"r{i}" for i in range(n_ref)]),
                          obsm={"X_emb": xrep})
        qry = MiniAnnData(X=np.zeros((nq, 2), np.float32), obs=pd.DataFrame({"grp": groups}, index=[f"q{i}" for i in range(nq)]),
                          obsm={"X_emb": yrep})
        cmap = cm_mod.CellMapper(qry, ref)
        cmap.compute_neighbors(n_neighbors=k, use_rep="X_emb", method="sklearn", only_yx=True)
        cmap.estimate_presence_score(groupby="grp", key_added="ps", **kw)
        out[f"all_{tag}"] = ref.obs["ps"].to_numpy()
        df = ref.obsm["ps"]
        out[f"groups_{tag}"] = df.to_numpy()
        out["group_names"] = np.array([str(c) for c in df.columns])
        if tag == "default":
            conn = cmap.knn.yx.knn_graph_connectivities()
            out["raw_all"] = np.asarray(conn.sum(axis=0)).ravel()
            out["yx_dist"], out["yx_idx"] = cmap.knn.yx.distances, cmap.knn.yx.indices
    return out


def expr_eval_case():
    """Per-gene agreement metrics exactly as EvaluationMixin.evaluate_expression_transfer computes them
    (evaluate.py:271-289: metric_func per gene over a cell mask, float32 array), using the reference module's own
    helpers (_rmse_zscore, _jensen_shannon_divergence; pearsonr / spearmanr as imported there), overall and per group."""
    import warnings

    ev = importlib.import_module("cellmapper.model.evaluate")
    rng = np.random.default_rng(13)
    n, G = 600, 40
    lam = rng.gamma(2.0, 1.5, G)
    orig = rng.poisson(lam, (n, G)).astype(np.float32)
    imputed = (0.6 * orig + 0.4 * lam + 0.5 * rng.standard_normal((n, G))).astype(np.float32)
    orig[:, 3] = 0.0                      # gene never measured: JS nan, pearson nan
    imputed[:, 5] = 1.25                  # constant imputation: pearson nan, z-score std -> 1
    orig[:, 7] = 2.0                      # constant measurement
    imputed[:, 9] = -np.abs(imputed[:, 9])  # all negative: clipped to zero mass -> JS nan
    labels = np.array(["t%d" % v for v in rng.integers(0, 3, n)])
    funcs = {
        "pearson": lambda a, b: ev.pearsonr(a, b)[0],
        "spearman": lambda a, b: ev.spearmanr(a, b)[0],
        "rmse": ev._rmse_zscore,
        "js": ev._jensen_shannon_divergence,
    }
    groups = list(pd.Series(labels).unique())
    out = dict(orig=orig, imputed=imputed, labels=labels, group_names=np.array(groups), **VERSIONS)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, f in funcs.items():
            def compute(mask, f=f):
                return np.array([f(orig[mask, i], imputed[mask, i]) for i in range(G)], dtype=np.float32)
            out[f"{name}_all"] = compute(np.ones(n, bool))
            out[f"{name}_groups"] = np.stack([compute(labels == g) for g in groups], axis=1)   # [G, n_groups]
    return out


def joint_embedding_case(utils_mod):
    """The reference's own truncated_svd_cross_covariance (utils.py:230-398; ARPACK on the implicit operator) on a dense
    float32 pair and on a CSR pair, and -- for joint PCA, which upstream delegates to scanpy (absent here) -- sklearn's
    PCA(svd_solver="arpack"), the estimator scanpy.pp.pca runs by default, on the concatenated matrices."""
    from sklearn.decomposition import PCA

    rng = np.random.default_rng(2024)
    nq, nr, G, nc = 300, 420, 96, 10
    # low-rank structure + noise, non-negative like expression data
    Z = rng.standard_normal((nq + nr, 12))
    W = rng.standard_normal((12, G))
    A = np.log1p(np.exp(Z @ W + 0.3 * rng.standard_normal((nq + nr, G)))).astype(np.float32)
    Xq, Xr = A[:nq], A[nq:]
    out = dict(Xq=Xq, Xr=Xr, n_comps=np.int64(nc), **VERSIONS)
    U, s, Vt = utils_mod.truncated_svd_cross_covariance(Xq, Xr, n_comps=nc, zero_center=True, implicit=True, random_state=0)
    out.update(dense_U=U, dense_s=s, dense_Vt=Vt)
    U, s, Vt = utils_mod.truncated_svd_cross_covariance(Xq, Xr, n_comps=nc, zero_center=False, implicit=True, random_state=0)
    out.update(nocenter_U=U, nocenter_s=s, nocenter_Vt=Vt)
    S = A * (rng.random(A.shape) < 0.25)
    Sq, Sr = sp.csr_matrix(S[:nq]), sp.csr_matrix(S[nq:])
    U, s, Vt = utils_mod.truncated_svd_cross_covariance(Sq, Sr, n_comps=nc, zero_center=True, implicit=True, random_state=0)
    out.update(**csr_parts("Sq", Sq), **csr_parts("Sr", Sr), sparse_U=U, sparse_s=s, sparse_Vt=Vt)
    joint = np.concatenate([Xr, Xq], axis=0)   # reference first, as embedding.py:35 concatenates
    out["pca_scores"] = PCA(n_components=nc, svd_solver="arpack", random_state=0).fit_transform(joint)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    cm_mod, knn_mod, utils_mod = load_reference()
    saved = {}
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "joint_embedding":
        np.savez_compressed(os.path.join(OUT, "joint_embedding.npz"), **joint_embedding_case(utils_mod))
        print("joint_embedding.npz written")
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "metrics":
        # metric pass-through (knn.py:343,429 hands `metric` to sklearn): manhattan and chebyshev, brute force (d = 50,
        # float32 and float64) and kd_tree (d = 2 self-mapping, d = 10 two-sample)
        def synth_(seed, n_ref, nq, dd, dtype):
            rng_ = np.random.default_rng(seed)
            return rng_.standard_normal((n_ref, dd)).astype(dtype), rng_.standard_normal((nq, dd)).astype(dtype)
        xr, yr = synth_(20, 1536, 192, 50, np.float32)
        xr64, yr64 = synth_(21, 1024, 128, 50, np.float64)
        xr10, yr10 = synth_(22, 1500, 100, 10, np.float32)
        coords = np.random.default_rng(23).uniform(0, 2000, (1500, 2)).astype(np.float32)
        xrz, yrz = xr.copy(), yr.copy()   # (sparse-ish rows: a fifth of the coordinates exactly zero)
        rz = np.random.default_rng(24)
        xrz[rz.random(xrz.shape) < 0.2] = 0.0
        yrz[rz.random(yrz.shape) < 0.2] = 0.0
        rb_ = np.random.default_rng(25)
        xb = (rb_.random((1200, 40)) < 0.33).astype(np.float32) * rb_.uniform(0.5, 3.0, (1200, 40)).astype(np.float32)
        yb = (rb_.random((150, 40)) < 0.33).astype(np.float32) * rb_.uniform(0.5, 3.0, (150, 40)).astype(np.float32)
        xh = rb_.integers(0, 3, (1200, 40)).astype(np.float32)
        yh = rb_.integers(0, 3, (150, 40)).astype(np.float32)
        rl_ = np.random.default_rng(26)
        ll64 = np.c_[rl_.uniform(-np.pi / 2, np.pi / 2, 1400), rl_.uniform(-np.pi, np.pi, 1400)]
        lq64 = np.c_[rl_.uniform(-np.pi / 2, np.pi / 2, 120), rl_.uniform(-np.pi, np.pi, 120)]
        lq64[:10] = ll64[:10] + 1e-4          # near neighbours: tiny angles
        lq64[10] = ll64[700]                  # an exact hit
        rn_ = np.random.default_rng(27)
        xn32, yn32 = synth_(28, 1200, 150, 30, np.float32)
        xn32[rn_.random(xn32.shape) < 0.06] = np.nan
        yn32[rn_.random(yn32.shape) < 0.06] = np.nan
        xn10, yn10 = synth_(29, 900, 90, 10, np.float32)
        xn10[rn_.random(xn10.shape) < 0.15] = np.nan
        yn10[rn_.random(yn10.shape) < 0.15] = np.nan
        yn10[3, :] = np.nan                   # a query with no coordinate at all: every distance NaN
        cases = {
            "synth_f32_manhattan_k15": (xr, yr, 15, "manhattan", "X_emb"),
            "synth_f64_manhattan_k30": (xr64, yr64, 30, "manhattan", "X_emb"),
            "synth_f32_chebyshev_k15": (xr, yr, 15, "chebyshev", "X_emb"),
            "synth_f64_chebyshev_k15": (xr64, yr64, 15, "chebyshev", "X_emb"),
            "synth_d10_manhattan_k15": (xr10, yr10, 15, "manhattan", "X_emb"),
            "spatial32_self_manhattan_k20": (coords, coords, 20, "manhattan", "spatial"),
            "spatial32_self_chebyshev_k20": (coords, coords, 20, "chebyshev", "spatial"),
            # round 3: sqeuclidean (brute force whatever d is; squared float64 distances out)
            "synth_f32_sqeuclidean_k15": (xr, yr, 15, "sqeuclidean", "X_emb"),
            "synth_d10_sqeuclidean_k15": (xr10, yr10, 15, "sqeuclidean", "X_emb"),
            "spatial32_self_sqeuclidean_k20": (coords, coords, 20, "sqeuclidean", "spatial"),
            # round 3: correlation (scipy.spatial.distance.cdist through sklearn's brute force; float64 distances out)
            "synth_f32_correlation_k15": (xr, yr, 15, "correlation", "X_emb"),
            "synth_f64_correlation_k30": (xr64, yr64, 30, "correlation", "X_emb"),
            "synth_d10_correlation_k15": (xr10, yr10, 15, "correlation", "X_emb"),
            # round 4: canberra and braycurtis (sklearn's own DistanceMetric: brute force / ArgKmin at d > 15, ball_tree below;
            # rows with zeros and sign changes so that the 0 / 0 rule and |x| + |y| are exercised)
            "synth_f32_canberra_k15": (xrz, yrz, 15, "canberra", "X_emb"),
            "synth_f64_canberra_k30": (xr64, yr64, 30, "canberra", "X_emb"),
            "synth_d10_canberra_k15": (xr10, yr10, 15, "canberra", "X_emb"),
            "synth_f32_braycurtis_k15": (xrz, yrz, 15, "braycurtis", "X_emb"),
            "synth_f64_braycurtis_k30": (xr64, yr64, 30, "braycurtis", "X_emb"),
            "synth_d10_braycurtis_k15": (xr10, yr10, 15, "braycurtis", "X_emb"),
            # round 4 (late): the counting metrics (scipy cdist on boolean copies under brute force, DistanceMetric on a ball
            # tree at d <= 15).  Binary-ish rows (a third non-zero; hamming: small integer values so that coordinates can agree)
            "bin40_hamming_k15": (xh, yh, 15, "hamming", "X_emb"),
            "bin10_hamming_k10": (xh[:, :10].copy(), yh[:, :10].copy(), 10, "hamming", "X_emb"),
            "bin40_jaccard_k15": (xb, yb, 15, "jaccard", "X_emb"),
            "bin10_jaccard_k10": (xb[:, :10].copy(), yb[:, :10].copy(), 10, "jaccard", "X_emb"),
            "bin40_dice_k15": (xb, yb, 15, "dice", "X_emb"),
            "bin40_rogerstanimoto_k15": (xb, yb, 15, "rogerstanimoto", "X_emb"),
            "bin40_russellrao_k15": (xb, yb, 15, "russellrao", "X_emb"),
            "bin40_sokalmichener_k15": (xb, yb, 15, "sokalmichener", "X_emb"),
            "bin40_sokalsneath_k15": (xb, yb, 15, "sokalsneath", "X_emb"),
            "bin10_sokalsneath_k10": (xb[:, :10].copy(), yb[:, :10].copy(), 10, "sokalsneath", "X_emb"),
            "bin40_yule_k15": (xb, yb, 15, "yule", "X_emb"),
            # (no d = 10 case for yule: no tree has the metric -- brute force at every width -- and at ten columns most of the k
            #  nearest distances are exactly 0, which the reference's gaussian kernel turns into NaN weights)
            # round 5: haversine (latitude, longitude in radians; two columns only).  sklearn: ball_tree on float64 copies when
            # k < n_ref / 2, ArgKmin brute force over HaversineDistance / HaversineDistance32 otherwise
            "latlon64_haversine_k12": (ll64, lq64, 12, "haversine", "spatial"),
            "latlon32_haversine_k12": (ll64.astype(np.float32), lq64.astype(np.float32), 12, "haversine", "spatial"),
            "latlon64_haversine_brute_k40": (ll64[:64].copy(), lq64[:32].copy(), 40, "haversine", "spatial", 4),
            "latlon32_haversine_brute_k40": (ll64[:64].astype(np.float32), lq64[:32].astype(np.float32), 40, "haversine", "spatial", 4),
            # round 5: nan_euclidean (sklearn.metrics.pairwise.nan_euclidean_distances through brute force; NaN = missing)
            "synth_f32_nan_euclidean_k15": (xn32, yn32, 15, "nan_euclidean", "X_emb"),
            "synth_f64_nan_euclidean_k15": (xn32.astype(np.float64), yn32.astype(np.float64), 15, "nan_euclidean", "X_emb"),
            "synth_d10_nan_euclidean_k10": (xn10, yn10, 10, "nan_euclidean", "X_emb"),
        }
        only = sys.argv[3] if len(sys.argv) > 3 else None   # e.g. `--only metrics sqeuclidean`: just the cases of one metric
        for name, case_ in cases.items():
            a_, b_, k_, m_, key_ = case_[:5]
            ncat_ = case_[5] if len(case_) > 5 else 20   # (tiny references: few label classes, so that every class is observed)
            if only and m_ != only:
                continue
            np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **run_mapping_case(cm_mod, name, a_, b_, k_, metric=m_, rep_key=key_, G=24, n_cat=ncat_))
            print(name, "written")
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "expr_eval":
        np.savez_compressed(os.path.join(OUT, "expr_eval.npz"), **expr_eval_case())
        print("expr_eval.npz written")
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "presence":
        np.savez_compressed(os.path.join(OUT, "presence.npz"), **presence_case(cm_mod))
        print("presence.npz written")
        return

    def save(name, d):
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **d)
        saved[name] = os.path.getsize(path)

    save("joint_embedding", joint_embedding_case(utils_mod))

    # ---- 1. small_data fixture of the reference's tests (tests/conftest.py:23-28), k=3, full mode
    x = np.array([[0, 0], [1, 0], [0, 1], [1, 1], [2, 2]], dtype=np.float64)
    nb = knn_mod.Neighbors(x, x.copy())
    nb.compute_neighbors(n_neighbors=3, method="sklearn", only_yx=False)
    d = dict(xrep=x, yrep=x.copy(), k=np.int64(3), **VERSIONS)
    for g in ("xx", "yy", "xy", "yx"):
        d[f"{g}_dist"] = getattr(nb, g).distances
        d[f"{g}_idx"] = getattr(nb, g).indices
    for kk in KERNELS:
        d.update(csr_parts(f"conn_{kk}", nb.yx.knn_graph_connectivities(kernel=kk)))
    d.update(csr_parts("bool_adj", nb.yx.boolean_adjacency()))
    d.update(csr_parts("graph_dist", nb.yx.knn_graph_distances))
    save("small_data", d)

    # ---- 2. sample_distances / sample_indices (tests/conftest.py:11-20) through every kernel + normalisation
    sd = np.array([[0.0, 1.0], [0.0, 2.0], [0.0, 3.0]], dtype=np.float64)
    si = np.array([[0, 1], [1, 2], [2, 0]], dtype=np.int64)
    nr = knn_mod.NeighborsResults(distances=sd, indices=si)
    d = dict(distances=sd, indices=si, **VERSIONS)
    q = MiniAnnData(X=np.zeros((3, 1), np.float32))
    for kk in KERNELS:
        conn = nr.knn_graph_connectivities(kernel=kk)
        d.update(csr_parts(f"conn_{kk}", conn))
        cmap = cm_mod.CellMapper(q)
        cmap.mapping_matrix = conn
        d.update(csr_parts(f"map_{kk}", cmap.mapping_matrix))
    d.update(csr_parts("conn_inverse_distance_eps", nr.knn_graph_connectivities(kernel="inverse_distance", epsilon=0.5)))
    save("sample_graph", d)

    # ---- 3. synthetic embeddings
    def synth(seed, n_ref, nq, dd, dtype):
        rng = np.random.default_rng(seed)
        return rng.standard_normal((n_ref, dd)).astype(dtype), rng.standard_normal((nq, dd)).astype(dtype)

    xr, yr = synth(0, 1536, 192, 50, np.float32)
    save("synth_f32_k15", run_mapping_case(cm_mod, "synth_f32_k15", xr, yr, 15, transfer_kernels=("gaussian", "inverse_distance")))
    save("synth_f32_k30", run_mapping_case(cm_mod, "synth_f32_k30", xr, yr, 30, transfer_kernels=("gaussian", "scarches", "equal")))
    xr64, yr64 = synth(1, 1024, 128, 50, np.float64)
    save("synth_f64_k30", run_mapping_case(cm_mod, "synth_f64_k30", xr64, yr64, 30))
    save("synth_f32_cosine_k15", run_mapping_case(cm_mod, "synth_f32_cosine_k15", xr, yr, 15, metric="cosine"))
    save("synth_f64_cosine_k15", run_mapping_case(cm_mod, "synth_f64_cosine_k15", xr64, yr64, 15, metric="cosine"))
    # wider embedding, d not multiple of anything, k=1 (identity-style), full mode incl. jaccard / hnoca
    xr2, yr2 = synth(2, 1024, 200, 37, np.float32)
    save("synth_full_d37_k10", run_mapping_case(cm_mod, "synth_full_d37_k10", xr2, yr2, 10, only_yx=False))
    # clustered (gaussian mixture) data with big offsets: stresses cancellation in the ||x||^2-2xy+||y||^2 form
    rng = np.random.default_rng(3)
    centers = rng.standard_normal((8, 50)) * 20.0
    xr3 = (centers[rng.integers(0, 8, 1536)] + rng.standard_normal((1536, 50))).astype(np.float32)
    yr3 = (centers[rng.integers(0, 8, 192)] + rng.standard_normal((192, 50))).astype(np.float32)
    save("synth_clustered_k30", run_mapping_case(cm_mod, "synth_clustered_k30", xr3, yr3, 30))
    # n_comps truncation: rep has 64 dims, reference uses first 50 (cellmapper.py:246-248)
    xr4, yr4 = synth(4, 1024, 128, 64, np.float32)
    save("synth_d64_trunc50_k15", run_mapping_case(cm_mod, "synth_d64_trunc50_k15", xr4, yr4, 15))

    # ---- 4. duplicated reference rows (exact ties in distance)
    xr5, yr5 = synth(5, 512, 64, 50, np.float32)
    xr5[100:140] = xr5[100]  # 40 identical rows > k
    xr5[300:305] = xr5[300]
    yr5[0] = xr5[100]
    yr5[1] = xr5[300]
    save("synth_dups_k15", run_mapping_case(cm_mod, "synth_dups_k15", xr5, yr5, 15))

    # ---- 5. spatial self-mapping, d=2 (sklearn picks kd_tree: neighbors/_base.py:622-655), k=90
    rng = np.random.default_rng(6)
    coords = rng.uniform(0, 2000, (1000, 2))
    out = run_mapping_case(cm_mod, "spatial_self_k90", coords, coords, 90, rep_key="spatial", G=24)
    save("spatial_self_k90", out)
    coords32 = rng.uniform(0, 2000, (2000, 2)).astype(np.float32)
    save("spatial32_self_k20", run_mapping_case(cm_mod, "spatial32_self_k20", coords32, coords32, 20, rep_key="spatial", G=24))
    # d=10 (still kd_tree) two-sample
    xr6, yr6 = synth(7, 1500, 100, 10, np.float32)
    save("synth_d10_k15", run_mapping_case(cm_mod, "synth_d10_k15", xr6, yr6, 15))

    # ---- 6. from_distances-style padded input (-1 / inf), via the reference's own ingest (knn.py:296-337)
    n = 40
    rng = np.random.default_rng(8)
    rows, cols, vals = [], [], []
    for i in range(n):
        deg = int(rng.integers(1, 7))
        nbrs = rng.choice(np.delete(np.arange(n), i), size=deg, replace=False)
        for j in nbrs:
            rows.append(i), cols.append(int(j)), vals.append(float(rng.uniform(0.1, 3.0)))
    dm = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    d = dict(**csr_parts("dm", dm), **VERSIONS)
    for inc, tag in ((None, "none"), (True, "true"), (False, "false")):
        nbf = knn_mod.Neighbors.from_distances(dm, include_self=inc)
        d[f"idx_{tag}"] = nbf.xx.indices
        d[f"dist_{tag}"] = nbf.xx.distances
        for kk in KERNELS:
            conn = nbf.xx.knn_graph_connectivities(kernel=kk)
            d.update(csr_parts(f"conn_{kk}_{tag}", conn))
            cmap = cm_mod.CellMapper(MiniAnnData(X=np.zeros((n, 1), np.float32)))
            cmap.mapping_matrix = conn
            d.update(csr_parts(f"map_{kk}_{tag}", cmap.mapping_matrix))
    save("padded_graph", d)

    # ---- 7. k = 1 identity mapping (tests/model/test_query_to_reference_mapping.py:75-92 restated on synthetic data)
    xr7, _ = synth(9, 300, 1, 20, np.float32)
    save("identity_k1", run_mapping_case(cm_mod, "identity_k1", xr7, xr7.copy(), 1, only_yx=False, n_cat=7, G=16))

    save("presence", presence_case(cm_mod))
    save("expr_eval", expr_eval_case())

    for k_, v in saved.items():
        print(f"{k_:28s} {v/1024:8.1f} KiB")
    print("total %.2f MiB" % (sum(saved.values()) / 2**20))


if __name__ == "__main__":
    main()
