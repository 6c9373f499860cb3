"""CPU-only tests: the C-ABI library loads and exports every symbol include/cellmap.h declares; host-side logic of
the drop-in boundary (container glue, argument checking, error conventions) behaves like the reference."""
import os
import re

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

from parity import load_golden, golden_csr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cellmap.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from cellmapper_amd import _native

    lib = _native.load_library()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/cellmap.h but not exported"
    assert set(_native.EXPORTS) <= set(syms)
    assert lib.cm_version() >= 100


def test_every_export_declares_argtypes():
    """ctypes marshals undeclared integer arguments as 32-bit C int: every bound symbol must carry argtypes whose length
    matches the parameter count of its declaration in include/cellmap.h."""
    from cellmapper_amd import _native

    lib = _native.load_library()
    src = open(os.path.join(ROOT, "include", "cellmap.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name in _native.EXPORTS:
        fn = getattr(lib, name)
        assert fn.argtypes is not None, f"{name} has no ctypes argtypes"
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, src, flags=re.S)
        assert m, f"{name} not declared in include/cellmap.h"
        params = m.group(1).strip()
        n_params = 0 if params in ("", "void") else params.count(",") + 1
        assert len(fn.argtypes) == n_params, f"{name}: {len(fn.argtypes)} argtypes for {n_params} declared parameters"
    assert set(declared_symbols()) == set(_native.EXPORTS), "EXPORTS must list exactly what the header declares"


def test_no_cpu_fallback_without_gpu():
    """Without a GPU every compute entry point must fail loudly (RuntimeError + build hint)."""
    from cellmapper_amd import _native
    from cellmapper_amd.check import check_deps

    lib = _native.load_library()
    if lib.cm_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="HIP backend"):
        _native.Handle(0)
    with pytest.raises(RuntimeError):
        check_deps("hip")
    with pytest.raises(RuntimeError, match="not registered"):
        check_deps("definitely_not_a_backend")
    from cellmapper_amd import Neighbors

    x = np.zeros((5, 20), np.float32)
    with pytest.raises(RuntimeError):
        Neighbors(x, x).compute_neighbors(n_neighbors=2)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cellmapper_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("CPU oracle restates", "").replace("oracle/cm_oracle.c", "").replace(
                    "the CPU oracle", "") or f in ("knn_kernels.hip",), f"{f} references the oracle"
                assert "import sklearn" not in txt and "from sklearn" not in txt, f"{f} imports sklearn"


def test_neighbors_results_container():
    from cellmapper_amd import NeighborsResults

    d = np.array([[0.0, 1.0], [0.0, 2.0], [0.0, 3.0]])
    i = np.array([[0, 1], [1, 2], [2, 0]])
    nr = NeighborsResults(distances=d, indices=i)
    assert (nr.n_samples, nr.n_neighbors, nr.shape) == (3, 2, (3, 3))
    assert NeighborsResults(d, i, n_targets=7).shape == (3, 7)
    with pytest.raises(ValueError, match="same shape"):
        NeighborsResults(distances=d, indices=np.zeros((3, 3), np.int64))
    with pytest.raises(ValueError, match="Unknown kernel"):
        nr.knn_graph_connectivities(kernel="nope")
    assert nr._get_valid_entries_mask().all()


def test_neighbors_repr_and_errors():
    from cellmapper_amd import Neighbors

    x = np.zeros((5, 2))
    nb = Neighbors(x, x.copy())
    assert "xrep_shape=(5, 2)" in repr(nb) and "xx=False" in repr(nb) and "self_mapping=False" in repr(nb)
    assert Neighbors(x)._is_self_mapping
    with pytest.raises(ValueError, match="Unknown method"):
        nb.compute_neighbors(method="annoy")
    with pytest.raises(ValueError, match="must be computed"):
        nb.get_adjacency_matrices()


def test_extract_neighbors_argument_checks_need_no_gpu():
    """The argument checks of the precomputed-distance ingest mirror the reference (utils.py:160-166) and fire before any
    GPU call; the computation itself is cm_ingest_distances (GPU parity: tests/test_gpu_boundary.py)."""
    from cellmapper_amd.utils import extract_neighbors_from_distances, get_n_comps

    with pytest.raises(TypeError):
        extract_neighbors_from_distances(np.zeros((3, 3)))
    with pytest.raises(ValueError, match="Square"):
        extract_neighbors_from_distances(sp.csr_matrix(np.zeros((3, 4))))
    assert get_n_comps(None, 80) == 50 and get_n_comps(None, 30) == 30 and get_n_comps(10, 30) == 10


def make_pair(n_ref=12, nq=5, G=4):
    from cellmapper_amd import MiniAnnData

    rng = np.random.default_rng(0)
    ref = MiniAnnData(X=rng.random((n_ref, G), dtype=np.float32),
                      obs=pd.DataFrame({"label": pd.Categorical(list("ab") * (n_ref // 2)), "num": np.arange(n_ref, dtype=float)}),
                      obsm={"X_emb": rng.standard_normal((n_ref, 20)).astype(np.float32)})
    qry = MiniAnnData(X=np.zeros((nq, 2), np.float32), obsm={"X_emb": rng.standard_normal((nq, 20)).astype(np.float32)})
    return qry, ref


def test_cellmapper_prerequisite_errors():
    from cellmapper_amd import CellMapper

    qry, ref = make_pair()
    cm = CellMapper(qry, ref)
    assert "CellMapper(query=" in repr(cm) and not cm._is_self_mapping
    with pytest.raises(ValueError, match="Neighbors have not been computed"):
        cm.compute_mapping_matrix()
    for fn, arg in ((cm.map_obsm, "X_emb"), (cm.map_layers, "X"), (cm.map_obs, "label")):
        with pytest.raises(ValueError, match="Mapping matrix has not been computed"):
            fn(arg)
    with pytest.raises(ValueError, match="Unknown fallback_representation"):   # use_rep=None computes a joint representation
        cm.compute_neighbors(use_rep=None, fallback_representation="nope")
    with pytest.raises(ValueError, match="only available in self-mapping"):
        cm.load_precomputed_distances()
    self_cm = CellMapper(qry)
    assert self_cm.reference is qry and self_cm._is_self_mapping and "self-mapping" in repr(self_cm)
    cm.mapping_matrix = None
    assert cm.mapping_matrix is None


def test_query_imputed_setter_validation():
    from cellmapper_amd import CellMapper

    qry, ref = make_pair()
    cm = CellMapper(qry, ref)
    cm.query_imputed = np.ones((qry.n_obs, ref.n_vars), np.float32)
    assert cm.query_imputed.X.shape == (5, 4) and cm.query_imputed.obs is qry.obs and cm.query_imputed.var is ref.var
    cm.query_imputed = sp.csr_matrix(np.ones((5, 4), np.float32))
    cm.query_imputed = pd.DataFrame(np.ones((5, 4)))
    with pytest.raises(ValueError, match="shape mismatch"):
        cm.query_imputed = np.ones((5, 3))
    with pytest.raises(ValueError, match="rows"):
        cm.query_imputed = pd.DataFrame(np.ones((4, 4)))
    with pytest.raises(ValueError, match="columns"):
        cm.query_imputed = pd.DataFrame(np.ones((5, 3)))
    with pytest.raises(TypeError):
        cm.query_imputed = [[1, 2], [3, 4]]
    cm.query_imputed = None
    assert cm.query_imputed is None


def test_grid_2d_layout():
    from cellmapper_amd.sharded import grid_2d

    # default (round 5): as few reference shards as fit a GPU -- all parallelism on the query axis when the reference fits
    assert grid_2d(1) == (1, 1) and grid_2d(2) == (2, 1) and grid_2d(4) == (4, 1) and grid_2d(8) == (8, 1)
    assert grid_2d(8, None, ref_bytes=300 << 20) == (8, 1) and grid_2d(8, None, ref_bytes=100 << 30) == (2, 4)
    assert grid_2d(8, None, ref_bytes=1 << 40) == (1, 8) and grid_2d(6, None, ref_bytes=70 << 30) == (2, 3)
    assert grid_2d(8, 8) == (1, 8) and grid_2d(8, 4) == (2, 4) and grid_2d(8, 1) == (8, 1)
    assert grid_2d(3) == (3, 1) and grid_2d(6, 4) == (2, 3) and grid_2d(3, 3) == (1, 3)
    for w in range(1, 17):
        for r in (None, 1, 2, 3, 4, 8, 16):
            g, s = grid_2d(w, r)
            assert g * s == w and g >= 1 and s >= 1


def test_shard_bounds_partition():
    from cellmapper_amd.sharded import shard_bounds, slice_bounds

    for n in (0, 1, 7, 1_500_000, 12_000_001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
            s = [slice_bounds(n, w, r) for r in range(w)]
            assert s[0][0] == 0 and s[-1][1] == n and all(s[i][1] == s[i + 1][0] for i in range(w - 1))
