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

    def weights_vote(self, dist, idx, n_ref, kernel, param, codes, n_cat, epsilon):
        d, i = dist.numpy(), idx.numpy()
        p = float(param[0]) if param is not None else 0.0
        w = {"gaussian": lambda: np.exp(-(d * d) / (2 * p * p)), "scarches": lambda: np.exp(-d / p),
             "inverse_distance": lambda: 1.0 / (d + epsilon), "equal": lambda: np.ones_like(d)}[kernel]()
        indptr, cols, v32, _ = self.o.mapping_matrix(w, d, i, n_ref)
        code, conf = self.o.label_vote(indptr, cols, v32, codes.numpy(), n_cat)
        return self.t.from_numpy(code), self.t.from_numpy(conf)

    def empty_words(self, n, like):
        return self.t.empty((n,), dtype=self.t.int64)


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
        sharded_map_labels(None, None, None, 0, 10, None, 3, None, 2, metric="braycurtis")


@pytest.mark.parametrize("world,ref_shards,exchange,kernel,metric", [
    (2, None, "alltoall", "gaussian", "euclidean"),    # 1 query group x 2 reference shards (pure reference sharding)
    (2, None, "allgather", "scarches", "euclidean"),
    (4, 2, "alltoall", "gaussian", "euclidean"),       # 2 query groups x 2 reference shards: the groups' all-to-all inside the world's (split sizes), world-wide sigma
    (4, 2, "allgather", "gaussian", "euclidean"),      # ... and through real sub-groups (new_group)
    (3, 3, "alltoall", "scarches", "euclidean"),       # odd world, ragged slices
    (2, None, "alltoall", "gaussian", "cosine"),       # cosine: merged values are NOT square-rooted
    (3, 3, "allgather", "inverse_distance", "cosine"),
    (2, None, "alltoall", "gaussian", "manhattan"),    # metric pass-through: merged values are the distances themselves
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
