#!/usr/bin/env python3
"""Host tier of the layer transfers (what `CellMapper.map(layer_key=...)` calls): numpy payload in -> numpy product out through
`cm_graph_spmm_dense` (A2) and `cm_graph_spmm_csr_nnz` + `cm_graph_spmm_csr` (A3), PCIe included.  Prints the time, the
bytes that cross the link and the rate they cross it at."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import scipy.sparse as sp
    from cellmapper_amd import _native
    n_ref = int(os.environ.get("HLP_NREF", 200_000)); nq = int(os.environ.get("HLP_NQ", 200_000))
    G = int(os.environ.get("HLP_GENES", 8000)); k = 30
    rng = np.random.default_rng(0)
    ref = rng.standard_normal((n_ref, 50), dtype=np.float32); qry = rng.standard_normal((nq, 50), dtype=np.float32)
    X = rng.random((n_ref, G), dtype=np.float32)
    h = _native.Handle(0)
    xr, yr = _native.DeviceRep(h, ref), _native.DeviceRep(h, qry)
    g = _native.DeviceGraph(h, xr, yr, k)
    g.mapping("gaussian")
    out = {"n_ref": n_ref, "nq": nq, "genes": G, "xfer": os.environ.get("CELLMAP_XFER", "off")}
    for rep in range(3):
        t0 = time.perf_counter()
        Y = g.spmm_dense(X)
        dt = time.perf_counter() - t0
        out.setdefault("dense_s", []).append(dt)
    link = X.nbytes + Y.nbytes
    out["dense_link_GB"] = link / 1e9
    out["dense_GBps_over_the_link"] = link / min(out["dense_s"]) / 1e9
    out["dense_checksum"] = float(Y[::997].sum())
    if os.environ.get("HLP_SPARSE", "1") != "0":
        Gs = int(os.environ.get("HLP_SGENES", 30000))
        nnz_row = int(0.03 * Gs)
        cols = np.sort(rng.integers(0, Gs, (min(n_ref, 60000), nnz_row)).astype(np.int32), axis=1)
        n_s = cols.shape[0]
        keep = np.ones_like(cols, bool); keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
        indptr = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int64)
        Xs = sp.csr_matrix((rng.random(int(indptr[-1]), dtype=np.float32), cols[keep], indptr), shape=(n_s, Gs))
        xr2, yr2 = _native.DeviceRep(h, ref[:n_s]), _native.DeviceRep(h, qry[:min(nq, 20000)])
        g2 = _native.DeviceGraph(h, xr2, yr2, k)
        g2.mapping("gaussian")
        for rep in range(2):
            t0 = time.perf_counter()
            ip, cc, vv = g2.spmm_csr(Xs.indptr.astype(np.int64), Xs.indices, Xs.data, Gs)
            out.setdefault("sparse_s", []).append(time.perf_counter() - t0)
        out["sparse_out_GB"] = (cc.nbytes + vv.nbytes) / 1e9
        out["sparse_GBps_of_output"] = out["sparse_out_GB"] / min(out["sparse_s"])
    print(json.dumps(out))


main()
