"""Reference-sharded k-NN -> kernel -> label transfer across the GPUs of one node.

Layout (2-D): ``world = query groups x reference shards``, rank = group * shards + shard.  A rank holds the reference
rows ``[lo, hi)`` of its shard (contiguous split), the full label-code vector, and the query rows of its group
(pure reference sharding = one group = every rank holds ALL queries).  Per step, inside a reference-shard group:

1. local exact top-k of every query against the rank's shard, with GLOBAL reference indices (``idx_offset``)
   and squared float64 distances;
2. ONE exchange over RCCL/xGMI: an all-to-all by query slice -- rank j receives, from every rank, the local
   top-k of query slice j -- ``[world, nq/world, k]`` (8x less traffic than an all-gather of the full lists;
   ``exchange="allgather"`` is the all-gather variant of the north-star wording);
3. k-way merge by (distance, global index): the same tie rule as the single-GPU search, so indices stay exact;
4. the global kernel bandwidth needs the mean (and for scarches the variance) of ALL nq*k distances: a
   3-double all-reduce;
5. weights / normalisation / label vote on the rank's query slice (label codes replicated).

The compute calls are injected (``ops``) so that the choreography can be unit-tested on CPU with gloo; in
production ``ops`` is :class:`HipOps` (device tensors through the C ABI).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous balanced split of ``n`` rows: the first ``n % world`` ranks get one extra row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def slice_bounds(nq: int, world: int, rank: int):
    """Equal query slices (nq is padded up to a multiple of world by the caller)."""
    per = (nq + world - 1) // world
    lo = min(nq, rank * per)
    return lo, min(nq, lo + per)


class HipOps:
    """Compute + tensor ops on the GPU (torch tensors, libcellmap_hip kernels)."""

    def __init__(self, pipeline):
        self.p = pipeline
        self.t = pipeline.torch

    def local_knn(self, ref_shard, qry, k, metric, idx_offset):
        return self.p.knn(ref_shard, qry, k, metric=metric, idx_offset=idx_offset, squared=True)

    def merge(self, dparts, iparts, k):
        return self.p.merge(dparts.contiguous(), iparts.contiguous(), k, sqrt_out=True)

    def moments(self, dist, idx, center=None):
        return self.p.moments(dist, idx, center)

    def param(self, m3, kernel):
        return self.p.kernel_param(m3, kernel)

    def weights_vote(self, dist, idx, n_ref, kernel, param, codes, n_cat, epsilon):
        cols, vals, cnt = self.p.weights(dist, idx, n_ref, kernel, param, epsilon)
        indptr, oc, ov = self.p.to_csr(cols, vals, cnt)
        return self.p.label_vote(indptr, oc, ov, codes, n_cat)

    def empty_like_parts(self, world, n, k, like):
        return self.t.empty((world, n, k), dtype=like.dtype, device=like.device)

    def pad_rows(self, x, n):
        if x.shape[0] == n:
            return x
        pad = self.t.zeros((n - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        return self.t.cat([x, pad], 0)


def grid_2d(world: int, ref_shards: int | None = None):
    """(query groups, reference shards) with ``groups * shards == world``.

    The reference is always partitioned when there is more than one GPU; by default two-way, and the
    remaining parallelism goes to the QUERY axis: query groups need no exchange at all, and a rank's shard stays large
    enough for the filter's log-n per-query overhead to amortise.  ``ref_shards=world`` gives the
    pure reference sharding of the north-star wording.
    """
    if ref_shards is None:
        ref_shards = 2 if world % 2 == 0 else (world if world > 1 else 1)
    ref_shards = max(1, min(world, int(ref_shards)))
    while world % ref_shards:
        ref_shards -= 1
    return world // ref_shards, ref_shards


def make_groups(dist_mod, world: int, n_groups: int, ref_shards: int):
    """Process groups of the 2-D layout: rank = group * ref_shards + shard.  Every rank creates every group (required by
    torch.distributed) and gets back the one it belongs to (None when a group is the whole world or a single rank)."""
    if world == 1 or ref_shards == 1:
        return None
    if n_groups == 1:
        return None  # the exchange runs on the default (world) group
    mine = None
    rank = dist_mod.get_rank()
    for g in range(n_groups):
        ranks = list(range(g * ref_shards, (g + 1) * ref_shards))
        pg = dist_mod.new_group(ranks=ranks)
        if rank in ranks:
            mine = pg
    return mine


def sharded_map_labels(ops, dist_mod, ref_shard, shard_lo, n_ref_total, qry, k, codes, n_cat, metric="euclidean",
                       kernel="gaussian", epsilon=1e-8, exchange="alltoall", group=None, ref_world=None, ref_rank=None):
    """Run one sharded step.  Returns ``(code, conf, (q_lo, q_hi), dist, idx)`` for this rank's query slice.

    ``dist_mod`` is ``torch.distributed`` (already initialised); ``ops`` supplies compute (HipOps in production).
    ``qry`` holds the queries of this rank's QUERY GROUP; ``group`` / ``ref_world`` / ``ref_rank`` describe the
    reference-shard group the exchange runs in (default: the whole world = pure reference sharding).  The bandwidth
    statistics are always reduced over the whole world.  ``(q_lo, q_hi)`` is relative to ``qry``.
    """
    n_all = dist_mod.get_world_size()
    world = n_all if ref_world is None else ref_world
    rank = dist_mod.get_rank() if ref_rank is None else ref_rank
    nq = qry.shape[0]
    per = (nq + world - 1) // world
    nq_pad = per * world
    d2, gi = ops.local_knn(ref_shard, qry, k, metric, shard_lo)  # [nq, k] squared float64, global int64 indices
    if world == 1:
        dist, idx = ops.merge(d2[None], gi[None], k)
        q_lo, q_hi = 0, nq
    else:
        d2p, gip = ops.pad_rows(d2, nq_pad), ops.pad_rows(gi, nq_pad)
        if exchange == "alltoall":
            rd = ops.empty_like_parts(world, per, k, d2p)
            ri = ops.empty_like_parts(world, per, k, gip)
            dist_mod.all_to_all_single(rd.view(world * per, k), d2p.view(world * per, k), group=group)
            dist_mod.all_to_all_single(ri.view(world * per, k), gip.view(world * per, k), group=group)
        elif exchange == "allgather":
            gd = ops.empty_like_parts(world, nq_pad, k, d2p)
            gg = ops.empty_like_parts(world, nq_pad, k, gip)
            dist_mod.all_gather_into_tensor(gd.view(world * nq_pad, k), d2p, group=group)
            dist_mod.all_gather_into_tensor(gg.view(world * nq_pad, k), gip, group=group)
            rd = gd[:, rank * per:(rank + 1) * per]
            ri = gg[:, rank * per:(rank + 1) * per]
        else:
            raise ValueError(f"unknown exchange '{exchange}'")
        q_lo, q_hi = slice_bounds(nq, world, rank)
        n_mine = q_hi - q_lo
        dist, idx = ops.merge(rd[:, :max(n_mine, 0)], ri[:, :max(n_mine, 0)], k)
    # global bandwidth: all-reduce the moments of the merged distances
    param = None
    if kernel in ("gaussian", "scarches"):
        m = ops.moments(dist, idx)
        if n_all > 1:
            dist_mod.all_reduce(m)
        param = ops.param(m, "gaussian")  # mean distance
        if kernel == "scarches":
            m2 = ops.moments(dist, idx, center=param)
            if n_all > 1:
                dist_mod.all_reduce(m2)
            param = ops.param(m2, "scarches")
    code, conf = ops.weights_vote(dist, idx, n_ref_total, kernel, param, codes, n_cat, epsilon)
    return code, conf, (q_lo, q_hi), dist, idx
