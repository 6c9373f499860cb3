"""World-size-2 gloo test (CPU) of the multi-GPU choreography in cellmapper_amd/sharded.py: reference sharding with
global indices, the exchange (all-to-all by query slice, and the all-gather variant), the k-way merge rule and the
all-reduced kernel bandwidth.  Compute is injected from the CPU oracle -- this tests the HOST logic that surrounds
the HIP kernels, not the kernels (those are covered by the -m gpu tests)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleOps:
    """CPU stand-in for HipOps with the same interface (torch CPU tensors in/out)."""

    def __init__(self):
        import torch
        from oracle import oracle as orc

        self.t, self.o = torch, orc

    def local_knn_packed(self, ref_shard, qry, k, metric, idx_offset, nq_pad):
        if metric == "cosine":   # float64 inputs in the test: the oracle's cosine values are already float64
            d, i = self.o.knn_cosine(ref_shard.numpy(), qry.numpy(), k)
            v = d.astype(np.float64)
        elif metric in ("manhattan", "chebyshev"):
            # the form (brute force or kd-tree arithmetic) is that of the WHOLE problem; at d = 20 it is brute force either way
            d, i = self.o.knn_minkowski_family(ref_shard.numpy(), qry.numpy(), k, metric, form="gemm")
            v = d
        else:
            d, i = self.o.knn_euclidean(ref_shard.numpy(), qry.numpy(), k)
            v = d * d
        buf = np.zeros((2, nq_pad, k), np.int64)
        buf[0, :qry.shape[0]] = v.view(np.int64)
        buf[1, :qry.shape[0]] = i + idx_offset
        return self.t.from_numpy(buf)

    def merge_packed(self, buf, nparts, part_stride, dist_off, idx_off, nq, k, metric):
        w = buf.numpy().reshape(-1)
        d = np.stack([w[dist_off + p * part_stride: dist_off + p * part_stride + nq * k].view(np.float64).reshape(nq, k)
                      for p in range(nparts)], 1).reshape(nq, -1)
        i = np.stack([w[idx_off + p * part_stride: idx_off + p * part_stride + nq * k].reshape(nq, k)
                      for p in range(nparts)], 1).reshape(nq, -1)
        order = np.lexsort((i, d), axis=1)[:, :k]
        dd = np.take_along_axis(d, order, 1)
        return (self.t.from_numpy(np.sqrt(dd) if metric == "euclidean" else dd),
                self.t.from_numpy(np.take_along_axis(i, order, 1)))

    def moments(self, dist, idx, center=None):
        c = 0.0 if center is None else float(center[0])
        v = dist.numpy().ravel() - c
        return self.t.tensor([v.sum(), (v * v).sum(), float(v.size)], dtype=self.t.float64)

    def param(self, m3, kernel):
        m = m3.numpy()
        if kernel == "gaussian":
            return self.t.tensor([m[0] / m[2]], dtype=self.t.float64)
        return self.t.tensor([(2.0 / np.sqrt(m[1] / m[2])) ** 2], dtype=self.t.float64)

    def weights(self, dist, idx, n_ref, kernel, param, epsilon):
        d, i = dist.numpy(), idx.numpy()
        p = float(param[0]) if param is not None else 0.0
        w = {"gaussian": lambda: np.exp(-(d * d) / (2 * p * p)), "scarches": lambda: np.exp(-d / p),
             "inverse_distance": lambda: 1.0 / (d + epsilon), "equal": lambda: np.ones_like(d)}[kernel]()
        indptr, cols, v32, _ = self.o.mapping_matrix(w, d, i, n_ref)
        n, k = d.shape
        cnt = np.diff(indptr).astype(np.int32)
        ec, ev = np.zeros((n, k), np.int32), np.zeros((n, k), np.float32)
        for r in range(n):   # CSR -> ELL (valid entries first, columns ascending)
            ec[r, :cnt[r]] = cols[indptr[r]:indptr[r + 1]]
            ev[r, :cnt[r]] = v32[indptr[r]:indptr[r + 1]]
        return self.t.from_numpy(ec), self.t.from_numpy(ev), self.t.from_numpy(cnt)

    def to_csr(self, cols, vals, cnt):
        c, v, n = cols.numpy(), vals.numpy(), cnt.numpy()
        indptr = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
        keep = np.arange(c.shape[1])[None, :] < n[:, None]
        return self.t.from_numpy(indptr), self.t.from_numpy(c[keep]), self.t.from_numpy(v[keep])

    def vote(self, cols, vals, cnt, codes, n_cat):
        ip, c, v = self.to_csr(cols, vals, cnt)
        code, conf = self.o.label_vote(ip.numpy(), c.numpy(), v.numpy(), codes.numpy(), n_cat)
        return self.t.from_numpy(code), self.t.from_numpy(conf)

    def spmm_dense(self, csr, B, out=None):
        r = self.t.from_numpy(self.o.spmm_dense(csr[0].numpy(), csr[1].numpy(), csr[2].numpy(), B.numpy()))
        if out is not None:
            out.copy_(r)
            return out
        return r

    def spmm_csr(self, csr, b_csr, n_out):
        import scipy.sparse as sp

        B = sp.csr_matrix((b_csr[2].numpy(), b_csr[1].numpy(), b_csr[0].numpy()), shape=(len(b_csr[0]) - 1, n_out))
        dense = self.o.spmm_csr(csr[0].numpy(), csr[1].numpy(), csr[2].numpy(), B)
        nz = dense != 0   # (csr_matmat drops exact zeros)
        indptr = np.concatenate([[0], np.cumsum(nz.sum(1))]).astype(np.int64)
        return (self.t.from_numpy(indptr), self.t.from_numpy(np.nonzero(nz)[1].astype(np.int32)), self.t.from_numpy(dense[nz]))

    def empty_words(self, n, like):
        return self.t.empty((n,), dtype=self.t.int64)

    def empty_i32(self, n, like):
        return self.t.empty((n,), dtype=self.t.int32)

    def cat(self, parts):
        return self.t.cat(parts)

    def equal(self, a, b):
        return bool(self.t.equal(a, b))

    def flag(self, ok, like):
        return self.t.tensor([1 if ok else 0], dtype=self.t.int32)


def _worker(rank, world, port, exchange, kernel, out_dir, ref_shards=None, metric="euclidean"):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from cellmapper_amd.sharded import grid_2d, make_groups, shard_bounds, sharded_map_labels

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    n_ref, nq, d, k, n_cat = 3001, 203, 20, 12, 7
    dt = np.float64 if metric == "cosine" else np.float32
    ref = rng.standard_normal((n_ref, d)).astype(dt)
    if metric == "euclidean":   # (the oracle's cosine goes through BLAS, whose rounding depends on the shard shape)
        ref[10] = ref[2500]  # exact tie across the shard boundary
    qry = rng.standard_normal((nq, d)).astype(dt)
    qry[0] = ref[10]
    codes = rng.integers(0, n_cat, n_ref).astype(np.int32)
    n_groups, rshards = grid_2d(world, ref_shards)
    qgroup, ref_rank = rank // rshards, rank % rshards
    pg = make_groups(dist, world, n_groups, rshards, exchange)
    lo, hi = shard_bounds(n_ref, rshards, ref_rank)
    g_lo, g_hi = shard_bounds(nq, n_groups, qgroup)
    code, conf, (q_lo, q_hi), dd, ii = sharded_map_labels(
        OracleOps(), dist, torch.from_numpy(ref[lo:hi].copy()), lo, n_ref, torch.from_numpy(qry[g_lo:g_hi].copy()), k,
        torch.from_numpy(codes), n_cat, metric=metric, kernel=kernel, exchange=exchange, group=pg, ref_world=rshards,
        ref_rank=ref_rank)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), code=code.numpy(), conf=conf.numpy(), q_lo=g_lo + q_lo, q_hi=g_lo + q_hi,
             dist=dd.numpy(), idx=ii.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_rejects_unsupported_metric():
    from cellmapper_amd.sharded import sharded_map_labels

    with pytest.raises(ValueError, match="not supported"):
        sharded_map_labels(None, None, None, 0, 10, None, 3, None, 2, metric="mahalanobis")


@pytest.mark.parametrize("world,ref_shards,exchange,kernel,metric", [
    (2, 2, "alltoall", "gaussian", "euclidean"),       # 1 query group x 2 reference shards (pure reference sharding)
    (2, 2, "allgather", "scarches", "euclidean"),
    (2, None, "alltoall", "gaussian", "euclidean"),    # the default layout (round 5): query groups only, the reference replicated -- no exchange, world-wide sigma
    (4, None, "alltoall", "scarches", "euclidean"),
    (4, 2, "alltoall", "gaussian", "euclidean"),       # 2 query groups x 2 reference shards: the groups' all-to-all inside the world's (split sizes), world-wide sigma
    (4, 2, "allgather", "gaussian", "euclidean"),      # ... and through real sub-groups (new_group)
    (3, 3, "alltoall", "scarches", "euclidean"),       # odd world, ragged slices
    (2, 2, "alltoall", "gaussian", "cosine"),          # cosine: merged values are NOT square-rooted
    (3, 3, "allgather", "inverse_distance", "cosine"),
    (2, 2, "alltoall", "gaussian", "manhattan"),       # metric pass-through: merged values are the distances themselves
])
def test_sharded_choreography(world, ref_shards, exchange, kernel, metric, tmp_path, oracle):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000) + world * 7 + (0 if exchange == "alltoall" else 1) + (3 if metric == "cosine" else (5 if metric == "manhattan" else 0))
    mp.spawn(_worker, args=(world, port, exchange, kernel, str(tmp_path), ref_shards, metric), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    n_ref, nq, d, k, n_cat = 3001, 203, 20, 12, 7
    dt = np.float64 if metric == "cosine" else np.float32
    ref = rng.standard_normal((n_ref, d)).astype(dt)
    if metric == "euclidean":
        ref[10] = ref[2500]
    qry = rng.standard_normal((nq, d)).astype(dt)
    qry[0] = ref[10]
    codes = rng.integers(0, n_cat, n_ref).astype(np.int32)
    full = oracle.run_path(ref, qry, k, metric=metric, kernel=kernel, labels=codes, n_cat=n_cat)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert int(parts[0]["q_lo"]) == 0 and int(parts[-1]["q_hi"]) == nq
    assert all(int(parts[r]["q_hi"]) == int(parts[r + 1]["q_lo"]) for r in range(world - 1))
    idx = np.concatenate([p["idx"] for p in parts])
    dist_ = np.concatenate([p["dist"] for p in parts])
    assert np.array_equal(idx, full["idx"]), "merged global indices must equal the unsharded search"
    np.testing.assert_allclose(dist_, full["dist"], rtol=1e-12)
    if metric == "euclidean":
        assert idx[0, 0] == 10 and idx[0, 1] == 2500  # lower global index wins the cross-shard tie
    np.testing.assert_array_equal(np.concatenate([p["code"] for p in parts]), full["pred_code"])
    np.testing.assert_allclose(np.concatenate([p["conf"] for p in parts]), full["conf"], rtol=1e-6)


# ---- aggregation half of SURVEY.md 8e: replicated small payloads, gene-column-sharded layers -------------------------
def _problem():
    rng = np.random.default_rng(5)
    n_ref, nq, d, k, G = 1501, 157, 16, 9, 40
    ref = rng.standard_normal((n_ref, d)).astype(np.float32)
    qry = rng.standard_normal((nq, d)).astype(np.float32)
    obsm = rng.standard_normal((n_ref, 2)).astype(np.float32)
    X = rng.random((n_ref, G), dtype=np.float32)
    Xs = X * (rng.random((n_ref, G)) < 0.2)     # sparse layer, 20 % density
    return ref, qry, obsm, X, Xs.astype(np.float32), k, G


def _worker_layers(rank, world, port, out_dir, ref_shards, break_alltoall):
    sys.path.insert(0, ROOT)
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist

    from cellmapper_amd import sharded as sh

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref, qry, obsm, X, Xs, k, G = _problem()
    lay = sh.Layout(world, rank, qry.shape[0], ref_shards)
    pg = sh.make_groups(dist, world, lay.n_groups, lay.ref_shards, "alltoall")
    lo, hi = sh.shard_bounds(ref.shape[0], lay.ref_shards, lay.ref_rank)
    state = sh.ExchangeState("alltoall")
    dist_mod = dist
    if break_alltoall:   # an all-to-all that raises (what an unsupported split-size call would do): the run must fall back
        class Broken:
            def __getattr__(self, name):
                return getattr(dist, name)

            def all_to_all_single(self, *a, **kw):
                raise RuntimeError("all_to_all_single with split sizes is not supported here")
        dist_mod = Broken()
    ops = OracleOps()
    outs = []
    for _ in range(2):   # step 1 self-checks the exchange, step 2 runs what step 1 decided
        rows = sh.sharded_mapping_rows(ops, dist_mod, torch.from_numpy(ref[lo:hi].copy()), lo, ref.shape[0],
                                       torch.from_numpy(qry[lay.group_lo:lay.group_hi].copy()), k, exchange="alltoall", group=pg,
                                       ref_world=lay.ref_shards, ref_rank=lay.ref_rank, state=state)
        outs.append(rows)
    assert torch.equal(outs[0]["idx"], outs[1]["idx"]) and torch.equal(outs[0]["vals"], outs[1]["vals"])
    rows = outs[1]
    my_lo, my_hi = lay.my_slice
    assert (lay.group_lo + rows["bounds"][0], lay.group_lo + rows["bounds"][1]) == (my_lo, my_hi)
    o_obsm = sh.sharded_map_obsm(ops, rows, torch.from_numpy(obsm))                       # query-sliced
    full = sh.gather_mapping_rows(ops, dist, rows, lay)                                   # ONE all-gather
    g_lo, g_hi = sh.shard_bounds(G, world, rank)                                          # this rank's gene columns
    o_dense = sh.sharded_map_layer_dense(ops, full, torch.from_numpy(np.ascontiguousarray(X[:, g_lo:g_hi])))
    slab = sp.csr_matrix(Xs[:, g_lo:g_hi])
    slab.sort_indices()
    o_csr = sh.sharded_map_layer_csr(ops, full, (torch.from_numpy(slab.indptr.astype(np.int64)),
                                                 torch.from_numpy(slab.indices.astype(np.int32)),
                                                 torch.from_numpy(slab.data.astype(np.float32))), g_hi - g_lo)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), q_lo=my_lo, q_hi=my_hi, g_lo=g_lo, g_hi=g_hi, obsm=o_obsm.numpy(),
             dense=o_dense.numpy(), c_ptr=o_csr[0].numpy(), c_cols=o_csr[1].numpy(), c_vals=o_csr[2].numpy(),
             mode=state.mode, note=str(state.note), full_cols=full[0].numpy(), full_vals=full[1].numpy(), full_cnt=full[2].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ref_shards,break_alltoall", [
    (2, 2, False),       # 1 query group x 2 reference shards
    (2, None, False),    # default layout: 2 query groups x 1 shard
    (4, 2, False),       # 2 x 2: ragged query groups (157 rows), world-wide all-gather of the mapping-matrix rows
    (3, 3, False),       # odd world, ragged slices and gene slabs
    (4, 2, True),        # the all-to-all raises on the first step -> world all-gather fall-back, same results
])
def test_sharded_obsm_and_layers(world, ref_shards, break_alltoall, tmp_path, oracle):
    import scipy.sparse as sp
    import torch.multiprocessing as mp

    port = 31500 + (os.getpid() % 1500) + world * 11 + (5 if break_alltoall else 0)
    mp.spawn(_worker_layers, args=(world, port, str(tmp_path), ref_shards, break_alltoall), nprocs=world, join=True)
    ref, qry, obsm, X, Xs, k, G = _problem()
    full = oracle.run_path(ref, qry, k, obsm=obsm, layer=X)
    want_csr = oracle.spmm_csr(full["indptr"], full["cols"], full["vals"], sp.csr_matrix(Xs))
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert all(str(p["mode"]) == ("allgather_world" if break_alltoall else "alltoall") for p in parts)
    if break_alltoall:
        assert "not supported" in str(parts[0]["note"])
    # query-sliced outputs tile the query rows; column-sliced outputs tile the genes
    assert int(parts[0]["q_lo"]) == 0 and int(parts[-1]["q_hi"]) == qry.shape[0]
    np.testing.assert_array_equal(np.concatenate([p["obsm"] for p in parts]), full["obsm_pred"])
    np.testing.assert_array_equal(np.concatenate([p["dense"] for p in parts], axis=1), full["layer_pred"])
    for p in parts:   # every rank holds the SAME replicated mapping matrix = the unsharded one
        cnt = p["full_cnt"]
        keep = np.arange(k)[None, :] < cnt[:, None]
        np.testing.assert_array_equal(p["full_cols"][keep], full["cols"])
        np.testing.assert_array_equal(p["full_vals"][keep], full["vals"])
        got = sp.csr_matrix((p["c_vals"], p["c_cols"], p["c_ptr"]), shape=(qry.shape[0], int(p["g_hi"]) - int(p["g_lo"]))).toarray()
        np.testing.assert_array_equal(got, want_csr[:, int(p["g_lo"]):int(p["g_hi"])])
