"""GPU tests of the streaming aggregation tiers (SURVEY.md section 7 "wide aggregation outputs"; reference call sites
cellmapper.py:338,372-373): column passes of the A3 accumulator, query-tiled host tiers of A2 / A3 (outputs that do not
fit the device budget), the device-tier A3 entry points.  Everything is compared bit for bit with the CPU oracle's
restatement of scipy's csr_matvecs / csr_matmat."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from cellmapper_amd import _native

    return _native.default_handle()


def mapping(rng, nq, n_ref, k, ragged=False):
    cols = np.sort(rng.integers(0, n_ref, (nq, k)), axis=1).astype(np.int32)
    vals = rng.random((nq, k), dtype=np.float32)
    if not ragged:
        return np.arange(nq + 1, dtype=np.int64) * k, cols.ravel(), vals.ravel()
    lens = rng.integers(0, k + 1, nq)
    lens[0] = 0
    keep = np.arange(k)[None, :] < lens[:, None]
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), cols[keep], vals[keep]


def csr_from_dense_rows(dense):
    m = sp.csr_matrix(dense)   # drops exact zeros, ascending columns
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float32)


@pytest.mark.parametrize("n_out,ct", [(500, None), (500, 64), (2000, 252), (40000, None)])
def test_spgemm_column_passes_match_oracle(hip, oracle, monkeypatch, n_out, ct):
    """n_out > 36 864 columns (two passes of the LDS accumulator), and forced narrow passes: same CSR as one pass."""
    rng = np.random.default_rng(n_out)
    n_ref, nq, k = 900, 260, 9
    indptr, cols, vals = mapping(rng, nq, n_ref, k, ragged=True)
    B = sp.random(n_ref, n_out, density=0.02, format="csr", dtype=np.float32, random_state=rng)
    B.data[:5] = 0.0   # stored zeros in the payload: products that are exactly zero are dropped from the output
    want = csr_from_dense_rows(oracle.spmm_csr(indptr, cols, vals, B))
    if ct is not None:
        monkeypatch.setenv("CELLMAP_SPGEMM_CT", str(ct))
    got = hip.spmm_csr(indptr, cols, vals, B.indptr, B.indices, B.data, n_out)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


def test_spgemm_host_tier_streams_query_tiles(hip, oracle, monkeypatch):
    """Output larger than the device budget: query tiles, copied out while the next tile is computed."""
    rng = np.random.default_rng(77)
    n_ref, nq, k, n_out = 3000, 5000, 12, 3000
    indptr, cols, vals = mapping(rng, nq, n_ref, k)
    B = sp.random(n_ref, n_out, density=0.03, format="csr", dtype=np.float32, random_state=rng)
    want = csr_from_dense_rows(oracle.spmm_csr(indptr, cols, vals, B))
    monkeypatch.setenv("CELLMAP_SPGEMM_BUDGET_MB", "4")   # 4 MB -> tiles of 256 Ki entries: ~30 tiles
    got = hip.spmm_csr(indptr, cols, vals, B.indptr, B.indices, B.data, n_out)
    assert got[0][-1] > 4 * (4 << 20) // 16, "the test must need several tiles"
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)


def test_spmm_dense_host_tier_streams_query_tiles(hip, oracle, monkeypatch):
    """Dense payload: column slabs x query tiles with a two-tile ring; same bits as one pass."""
    rng = np.random.default_rng(78)
    n_ref, nq, k, G = 2500, 9000, 10, 640
    indptr, cols, vals = mapping(rng, nq, n_ref, k, ragged=True)
    for dt in (np.float32, np.float64):
        X = rng.random((n_ref, G)).astype(dt)
        want = oracle.spmm_dense(indptr, cols, vals, X)
        monkeypatch.setenv("CELLMAP_SPMM_BUDGET_MB", "8")   # forces 1024-row query tiles and several column slabs
        np.testing.assert_array_equal(hip.spmm_dense(indptr, cols, vals, X), want)
        monkeypatch.delenv("CELLMAP_SPMM_BUDGET_MB")
        np.testing.assert_array_equal(hip.spmm_dense(indptr, cols, vals, X), want)


def test_spgemm_device_tier_tiles_equal_the_whole(oracle):
    """cm_spmm_csr_nnz_dev / cm_spmm_csr_dev on query tiles (views of the indptr), with and without caller-provided
    worst-case buffers, against the oracle."""
    import torch

    from cellmapper_amd.device import DevicePipeline

    pipe = DevicePipeline(0)
    rng = np.random.default_rng(79)
    n_ref, nq, k, n_out = 1500, 1000, 8, 700
    indptr, cols, vals = mapping(rng, nq, n_ref, k, ragged=True)
    B = sp.random(n_ref, n_out, density=0.05, format="csr", dtype=np.float32, random_state=rng)
    want = csr_from_dense_rows(oracle.spmm_csr(indptr, cols, vals, B))
    dev = pipe.device
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    d = dict(indptr=t(indptr, np.int64), cols=t(cols, np.int32), vals=t(vals, np.float32), bp=t(B.indptr, np.int64),
             bc=t(B.indices, np.int32), bv=t(B.data, np.float32))
    with torch.cuda.stream(pipe.stream):
        tile = 300
        cap = tile * n_out
        buf = (torch.empty(tile + 1, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
               torch.empty(cap, dtype=torch.float32, device=dev))
        for use_buf in (False, True):
            ptrs, cs, vs = [np.zeros(1, np.int64)], [], []
            for q0 in range(0, nq, tile):
                q1 = min(nq, q0 + tile)
                optr, oc, ov = pipe.spmm_csr(d["indptr"], d["cols"], d["vals"], d["bp"], d["bc"], d["bv"], n_out, q0, q1,
                                             out=buf if use_buf else None)
                pipe.synchronize()
                op = optr[:q1 - q0 + 1].cpu().numpy()
                ptrs.append(op[1:] + ptrs[-1][-1])
                cs.append(oc[:op[-1]].cpu().numpy())
                vs.append(ov[:op[-1]].cpu().numpy())
            np.testing.assert_array_equal(np.concatenate(ptrs), want[0])
            np.testing.assert_array_equal(np.concatenate(cs), want[1])
            np.testing.assert_array_equal(np.concatenate(vs), want[2])


def test_spgemm_row_order_is_a_pure_permutation(oracle):
    """cm_locality_order_dev + cm_spmm_csr_*_dev with a row order: tile rows are the product rows of mapping-matrix rows
    order[q0:q1] -- the same bits as the unordered product, only arranged for L2 reuse; the order is a permutation that keeps
    rows with the same smallest neighbour column together."""
    import torch

    from cellmapper_amd.device import DevicePipeline

    pipe = DevicePipeline(0)
    rng = np.random.default_rng(80)
    n_ref, nq, k, n_out = 400, 3000, 6, 500
    indptr, cols, vals = mapping(rng, nq, n_ref, k, ragged=True)
    B = sp.random(n_ref, n_out, density=0.05, format="csr", dtype=np.float32, random_state=rng)
    dense = oracle.spmm_csr(indptr, cols, vals, B)
    dev = pipe.device
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    d = dict(indptr=t(indptr, np.int64), cols=t(cols, np.int32), vals=t(vals, np.float32), bp=t(B.indptr, np.int64),
             bc=t(B.indices, np.int32), bv=t(B.data, np.float32))
    with torch.cuda.stream(pipe.stream):
        order = pipe.locality_order(d["indptr"], d["cols"], n_ref)
        pipe.synchronize()
        o = order.cpu().numpy()
        assert np.array_equal(np.sort(o), np.arange(nq))
        first = np.where(np.diff(indptr) > 0, cols[np.minimum(indptr[:-1], len(cols) - 1)], n_ref)
        assert np.all(np.diff(first[o]) >= 0), "rows must be grouped by their smallest neighbour column"
        for q0 in range(0, nq, 1100):
            q1 = min(nq, q0 + 1100)
            optr, oc, ov = pipe.spmm_csr(d["indptr"], d["cols"], d["vals"], d["bp"], d["bc"], d["bv"], n_out, q0, q1, order=order)
            pipe.synchronize()
            op, occ, ovv = optr.cpu().numpy(), oc.cpu().numpy(), ov.cpu().numpy()
            for r in range(q1 - q0):
                row = dense[o[q0 + r]]
                wc = np.flatnonzero(row)
                assert np.array_equal(occ[op[r]:op[r + 1]], wc) and np.array_equal(ovv[op[r]:op[r + 1]], row[wc])


@pytest.mark.gpu
@pytest.mark.parametrize("with_order,with_index", [(False, False), (True, False), (False, True), (True, True)])
def test_spgemm_single_pass_equals_the_two_pass_pair(oracle, with_order, with_index):
    """cm_spmm_csr_fused_dev (one gather, offsets from the chained scan inside the kernel) against the oracle and the count +
    fill pair: identical indptr / columns / values, with ragged and EMPTY mapping rows, payload values of both signs (exact
    zeros from cancellation are dropped, csr_matmat's rule), thousands of rows (look-back over more than 64 predecessors
    and many rounds of tickets) and a capacity that is too small (counts stay exact, rows beyond it are not written).
    with_index: the wave-sliced kernel (cm_payload_index_dev), no barrier per neighbour."""
    import torch

    from cellmapper_amd.device import DevicePipeline

    pipe = DevicePipeline(0)
    rng = np.random.default_rng(81)
    n_ref, nq, k, n_out = 300, 7000, 5, 700
    indptr, cols, vals = mapping(rng, nq, n_ref, k, ragged=True)
    vals = np.full_like(vals, 0.25)                         # equal weights: +v and -v cancel exactly
    B = sp.random(n_ref, n_out, density=0.04, format="csr", dtype=np.float32, random_state=rng)
    B.data = rng.choice(np.array([-2.0, -1.0, 1.0, 2.0], dtype=np.float32), size=B.data.shape[0])
    dense = oracle.spmm_csr(indptr, cols, vals, B)
    Babs = B.copy()
    Babs.data = np.abs(Babs.data)
    assert ((dense == 0) & (oracle.spmm_csr(indptr, cols, vals, Babs) != 0)).any(), "the case needs exact cancellations"
    dev = pipe.device
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    d = dict(indptr=t(indptr, np.int64), cols=t(cols, np.int32), vals=t(vals, np.float32), bp=t(B.indptr, np.int64),
             bc=t(B.indices, np.int32), bv=t(B.data, np.float32))
    args = (d["indptr"], d["cols"], d["vals"], d["bp"], d["bc"], d["bv"], n_out)
    with torch.cuda.stream(pipe.stream):
        order = pipe.locality_order(d["indptr"], d["cols"], n_ref) if with_order else None
        pidx = pipe.payload_index(d["bp"], d["bc"], n_out) if with_index else None
        rows = order.cpu().numpy() if with_order else np.arange(nq)
        want_ptr = np.concatenate([[0], np.cumsum([(dense[r] != 0).sum() for r in rows])])
        cap = int(want_ptr[-1])
        for capacity in (cap + 17, cap, cap // 2):
            out = (torch.full((nq + 1,), -1, dtype=torch.int64, device=dev), torch.full((max(capacity, 1),), -7, dtype=torch.int32, device=dev),
                   torch.full((max(capacity, 1),), -7.0, dtype=torch.float32, device=dev))
            optr, oc, ov = pipe.spmm_csr(*args, out=out, order=order, single_pass=True, payload_index=pidx)
            pipe.synchronize()
            op, occ, ovv = optr.cpu().numpy(), oc.cpu().numpy(), ov.cpu().numpy()
            assert np.array_equal(op, want_ptr)
            n_ok = int(np.searchsorted(want_ptr, capacity, side="right")) - 1     # rows that end within the capacity
            for r in range(0, n_ok, 13):
                row = dense[rows[r]]
                wc = np.flatnonzero(row)
                assert np.array_equal(occ[op[r]:op[r + 1]], wc) and np.array_equal(ovv[op[r]:op[r + 1]], row[wc])
            assert np.all(occ[want_ptr[n_ok]:capacity] == -7), "rows beyond the capacity must not be written"
        # completion order: a permutation of the rows, each row's content unchanged
        out = (torch.empty(nq + 1, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
               torch.empty(cap, dtype=torch.float32, device=dev))
        ids = torch.full((nq,), -1, dtype=torch.int32, device=dev)
        optr, oc, ov = pipe.spmm_csr(*args, out=out, order=order, single_pass=True, out_rows=ids, payload_index=pidx)
        pipe.synchronize()
        op, occ, ovv, idn = optr.cpu().numpy(), oc.cpu().numpy(), ov.cpu().numpy(), ids.cpu().numpy()
        assert np.array_equal(np.sort(idn), np.arange(nq)) and op[0] == 0 and op[-1] == cap
        assert np.array_equal(np.diff(op), np.diff(want_ptr)[idn])
        for j in range(0, nq, 11):
            row = dense[rows[idn[j]]]
            wc = np.flatnonzero(row)
            assert np.array_equal(occ[op[j]:op[j + 1]], wc) and np.array_equal(ovv[op[j]:op[j + 1]], row[wc])
        # the two-pass pair on the same buffers gives the same bytes
        bufs = lambda: (torch.empty(nq + 1, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
                        torch.empty(cap, dtype=torch.float32, device=dev))
        out1, out2 = bufs(), bufs()
        pipe.spmm_csr(*args, out=out1, order=order, single_pass=True, payload_index=pidx)
        pipe.spmm_csr(*args, out=out2, order=order, single_pass=False)
        pipe.synchronize()
        for x, y in zip(out1, out2):
            assert torch.equal(x, y)


@pytest.mark.gpu
def test_spgemm_sliced_kernel_shapes(oracle):
    """The wave-sliced product on shapes that leave its fast paths: a slice of more than 128 entries per neighbour row (dense
    payload rows), mapping rows with more than 128 neighbours, fewer buckets than waves (narrow payloads), an output width
    that is no multiple of the bucket width; and the index builder's refusal of a payload that is not canonical CSR."""
    import torch

    from cellmapper_amd.device import DevicePipeline

    pipe = DevicePipeline(0)
    dev = pipe.device
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
    # (slices of ~225 entries: 4 per lane in registers; > 128 neighbours; 2 buckets for 8 waves; short slices; slices of ~130..250;
    #  slices of ~700 entries: 8 per lane + the scalar loop)
    # round 3: entries per lane are chosen by the payload's longest slice -- 1 (<= 64), 2 (<= 128), 3 (<= 192), 4 (<= 256) consecutive
    # entries per 4 / 8 / 12 / 16-byte load, 8 beyond -- and a vector load may run past the end of its slice: the LAST payload row,
    # whose last slice ends with the arrays, is given a short last slice and a place in every mapping row (the slice at the very end
    # of the arrays must take the entry-per-lane loop instead of reading past them)
    for seed, (n_ref, nq, k, n_out, density) in enumerate([(120, 300, 7, 3000, 0.6), (400, 150, 200, 900, 0.05), (60, 200, 4, 90, 0.3),
                                                            (80, 100, 3, 1000, 0.2), (100, 200, 5, 4000, 0.35), (50, 120, 4, 8000, 0.7),
                                                            (90, 150, 6, 4000, 0.17), (90, 150, 6, 4000, 0.29), (70, 130, 9, 2000, 0.1),
                                                            (64, 90, 5, 30000, 0.03), (64, 90, 5, 12000, 0.05), (64, 90, 5, 16000, 0.012)]):
        rng = np.random.default_rng(90 + seed)
        indptr, cols, vals = mapping(rng, nq, n_ref, k, ragged=True)
        B = sp.random(n_ref, n_out, density=density, format="lil", dtype=np.float32, random_state=rng)
        W = max(64, (((n_out + 7) // 8) + 63) // 64 * 64)
        last_lo = (n_out - 1) // W * W                      # first column of the last bucket
        B[n_ref - 1, last_lo:] = 0
        B[n_ref - 1, n_out - 1] = 1.5                       # the last row's last slice: ONE entry, the last of the arrays
        B = B.tocsr()
        B.eliminate_zeros()
        B.sort_indices()
        has = np.diff(indptr) > 0
        cols = cols.copy()
        cols[indptr[:-1][has]] = n_ref - 1                  # every non-empty mapping row gathers the last payload row
        dense = oracle.spmm_csr(indptr, cols, vals, B)
        want_ptr = np.concatenate([[0], np.cumsum((dense != 0).sum(1))])
        cap = int(want_ptr[-1])
        d = dict(indptr=t(indptr, np.int64), cols=t(cols, np.int32), vals=t(vals, np.float32), bp=t(B.indptr, np.int64),
                 bc=t(B.indices, np.int32), bv=t(B.data, np.float32))
        with torch.cuda.stream(pipe.stream):
            pidx = pipe.payload_index(d["bp"], d["bc"], n_out)
            out = (torch.empty(nq + 1, dtype=torch.int64, device=dev), torch.empty(max(cap, 1), dtype=torch.int32, device=dev),
                   torch.empty(max(cap, 1), dtype=torch.float32, device=dev))
            optr, oc, ov = pipe.spmm_csr(d["indptr"], d["cols"], d["vals"], d["bp"], d["bc"], d["bv"], n_out, out=out, single_pass=True,
                                         payload_index=pidx)
            pipe.synchronize()
        op, occ, ovv = optr.cpu().numpy(), oc.cpu().numpy(), ov.cpu().numpy()
        assert np.array_equal(op, want_ptr)
        for r in range(nq):
            wc = np.flatnonzero(dense[r])
            assert np.array_equal(occ[op[r]:op[r + 1]], wc) and np.array_equal(ovv[op[r]:op[r + 1]], dense[r][wc]), (seed, r)
    # not canonical: a row with descending columns
    bad_cols = B.indices.copy()
    lo, hi = B.indptr[1], B.indptr[2]
    assert hi - lo >= 2
    bad_cols[lo:hi] = bad_cols[lo:hi][::-1]
    with pytest.raises(ValueError, match="ascending"):
        pipe.payload_index(d["bp"], t(bad_cols, np.int32), n_out)


def test_host_tier_single_pass_product_equals_the_count_and_fill_pair(hip, oracle, monkeypatch):
    """Round 3: when a worst-case buffer for the whole product fits the device budget, cm_spmm_csr_nnz makes counts, offsets
    and entries in ONE gather and cm_spmm_csr only copies them out.  Same bits as the two-pass pair (forced here by a
    4 MB budget) and as the oracle; a non-canonical payload (unsorted columns) silently takes the pair."""
    rng = np.random.default_rng(12)
    n_ref, nq, k, G = 7000, 2600, 25, 900
    idx = np.stack([rng.choice(n_ref, k, replace=False) for _ in range(nq)]).astype(np.int64)
    dist = np.sort(rng.random((nq, k)) + 0.1, axis=1)
    ip, cc, vv, _, _ = hip.mapping_matrix(dist, idx, n_ref, "gaussian")
    X = sp.random(n_ref, G, density=0.08, format="csr", dtype=np.float32, random_state=rng)
    X.sort_indices()
    one = hip.spmm_csr(ip, cc, vv, X.indptr, X.indices, X.data, G)                       # single pass (fits easily)
    monkeypatch.setenv("CELLMAP_SPGEMM_BUDGET_MB", "4")
    two = hip.spmm_csr(ip, cc, vv, X.indptr, X.indices, X.data, G)                       # tiled count + fill pair
    monkeypatch.delenv("CELLMAP_SPGEMM_BUDGET_MB")
    for a_, b_ in zip(one, two):
        assert np.array_equal(a_, b_)
    want = oracle.spmm_csr(ip, cc, vv, X)
    got = sp.csr_matrix((one[2], one[1], one[0]), shape=(nq, G)).toarray()
    assert np.array_equal(got, want)
    # unsorted payload columns: the payload index refuses it, the barrier pair (which only needs distinct columns) runs
    Xu = X.copy()
    for r in range(0, n_ref, 3):
        a0, a1 = Xu.indptr[r], Xu.indptr[r + 1]
        if a1 - a0 > 1:
            Xu.indices[a0:a1] = Xu.indices[a0:a1][::-1].copy()
            Xu.data[a0:a1] = Xu.data[a0:a1][::-1].copy()
    un = hip.spmm_csr(ip, cc, vv, Xu.indptr, Xu.indices, Xu.data, G)
    assert np.array_equal(sp.csr_matrix((un[2], un[1], un[0]), shape=(nq, G)).toarray(), want)
