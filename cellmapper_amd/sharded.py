"""Reference-sharded k-NN -> kernel -> label transfer across the GPUs of one node.

Layout (2-D): ``world = query groups x reference shards``, rank = group * shards + shard.  A rank holds the reference
rows ``[lo, hi)`` of its shard (contiguous split), the full label-code vector, and the query rows of its group
(pure reference sharding = one group = every rank holds ALL queries).  Per step, inside a reference-shard group:

1. local exact top-k of every query against the rank's shard, with GLOBAL reference indices (``idx_offset``)
   and squared float64 distances;
2. ONE collective over RCCL/xGMI on a packed buffer (distances and indices travel together): an all-to-all by
   query slice -- rank j receives, from every rank, the local top-k of query slice j -- ``[world, 2, nq/world, k]``
   (8x less traffic than an all-gather of the full lists; ``exchange="allgather"`` is the all-gather variant of the
   north-star wording);
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

    def local_knn_packed(self, ref_shard, qry, k, metric, idx_offset, nq_pad):
        """Local exact top-k written straight into the PACKED exchange buffer ``[2, nq_pad, k]`` of 8-byte words:
        plane 0 = float64 ranking values (squared for euclidean; float64 cosine / manhattan / chebyshev distances), plane 1 = int64
        global indices.
        One buffer = one collective per exchange.  Padded rows are zero (never merged: slices are cut at nq)."""
        t = self.t
        nq = qry.shape[0]
        buf = t.zeros((2, nq_pad, k), dtype=t.int64, device=qry.device) if nq_pad != nq else \
            t.empty((2, nq_pad, k), dtype=t.int64, device=qry.device)
        self.p.knn(ref_shard, qry, k, metric=metric, idx_offset=idx_offset, squared=True, f64_out=True,
                   out=(buf[0].view(t.float64)[:nq], buf[1][:nq]))
        return buf

    def merge_packed(self, buf, nparts, part_stride, dist_off, idx_off, nq, k, metric):
        """k-way merge of ``nparts`` lists per query living in one packed buffer (offsets / stride in 8-byte words)."""
        return self.p.merge_strided(buf, nparts, part_stride, dist_off, idx_off, nq, k, sqrt_out=(metric == "euclidean"))

    def moments(self, dist, idx, center=None):
        return self.p.moments(dist, idx, center)

    def param(self, m3, kernel):
        return self.p.kernel_param(m3, kernel)

    def weights_vote(self, dist, idx, n_ref, kernel, param, codes, n_cat, epsilon):
        cols, vals, cnt = self.p.weights(dist, idx, n_ref, kernel, param, epsilon)
        indptr, oc, ov = self.p.to_csr(cols, vals, cnt)
        return self.p.label_vote(indptr, oc, ov, codes, n_cat)

    def empty_words(self, n, like):
        return self.t.empty((n,), dtype=self.t.int64, device=like.device)


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


class WorldSlice:
    """The reference-shard group [first, first + count) of a 2-D layout, addressed THROUGH THE WORLD communicator: the
    all-to-all exchange of a group is an ``all_to_all_single`` of the whole world whose split sizes are zero outside the
    group.  No sub-communicator is created (no ``new_group`` / communicator split): the only RCCL communicator of a run is
    the default one."""

    def __init__(self, first: int, count: int):
        self.first, self.count = int(first), int(count)


def make_groups(dist_mod, world: int, n_groups: int, ref_shards: int, exchange: str = "alltoall"):
    """Process groups of the 2-D layout: rank = group * ref_shards + shard.  For the all-to-all exchange a WorldSlice (see
    there); for the all-gather exchange a real sub-group: every rank creates every group (required by torch.distributed)
    and gets back the one it belongs to.  None when a group is the whole world or a single rank."""
    if world == 1 or ref_shards == 1:
        return None
    if n_groups == 1:
        return None  # the exchange runs on the default (world) group
    if exchange == "alltoall":
        return WorldSlice((dist_mod.get_rank() // ref_shards) * ref_shards, ref_shards)
    mine = None
    rank = dist_mod.get_rank()
    for g in range(n_groups):
        ranks = list(range(g * ref_shards, (g + 1) * ref_shards))
        pg = dist_mod.new_group(ranks=ranks)
        if rank in ranks:
            mine = pg
    return mine


SUPPORTED_METRICS = ("euclidean", "cosine", "manhattan", "chebyshev")


def sharded_map_labels(ops, dist_mod, ref_shard, shard_lo, n_ref_total, qry, k, codes, n_cat, metric="euclidean",
                       kernel="gaussian", epsilon=1e-8, exchange="alltoall", group=None, ref_world=None, ref_rank=None):
    """Run one sharded step.  Returns ``(code, conf, (q_lo, q_hi), dist, idx)`` for this rank's query slice.

    ``dist_mod`` is ``torch.distributed`` (already initialised); ``ops`` supplies compute (HipOps in production).
    ``qry`` holds the queries of this rank's QUERY GROUP; ``group`` / ``ref_world`` / ``ref_rank`` describe the
    reference-shard group the exchange runs in (default: the whole world = pure reference sharding).  The bandwidth
    statistics are always reduced over the whole world.  ``(q_lo, q_hi)`` is relative to ``qry``.

    The exchange is ONE collective on a packed buffer (ranking values and global indices travel together as 8-byte
    words): ``all_to_all_single`` of ``[world][2][per][k]`` (default), or ``all_gather_into_tensor`` of ``[2][nq_pad][k]``.
    Shard-local lists carry the float64 values the single-GPU search ranks by (squared euclidean; float64 cosine even
    for float32 input), so the merged order -- and therefore every index -- equals the unsharded result.
    """
    if metric not in SUPPORTED_METRICS:
        raise ValueError(f"Metric '{metric}' is not supported by the sharded hip backend (supported: euclidean, cosine, manhattan, chebyshev).")
    n_all = dist_mod.get_world_size()
    world = n_all if ref_world is None else ref_world
    rank = dist_mod.get_rank() if ref_rank is None else ref_rank
    nq = qry.shape[0]
    per = (nq + world - 1) // world
    nq_pad = per * world
    send = ops.local_knn_packed(ref_shard, qry, k, metric, shard_lo, nq_pad)   # [2, nq_pad, k] 8-byte words
    pk = per * k
    if world == 1:
        q_lo, q_hi = 0, nq
        dist, idx = ops.merge_packed(send, 1, 2 * nq_pad * k, 0, nq_pad * k, nq, k, metric)
    else:
        q_lo, q_hi = slice_bounds(nq, world, rank)
        n_mine = max(q_hi - q_lo, 0)
        if exchange == "alltoall":
            # chunk j of the send buffer = what rank j merges: [2][per][k] of query slice j
            packed = send.view(2, world, pk).transpose(0, 1).contiguous().view(-1)
            recv = ops.empty_words(world * 2 * pk, send)
            if isinstance(group, WorldSlice):   # the group's all-to-all inside the world's: empty chunks for everybody else
                splits = [2 * pk if group.first <= r < group.first + group.count else 0 for r in range(n_all)]
                dist_mod.all_to_all_single(recv, packed, output_split_sizes=splits, input_split_sizes=splits)
            else:
                dist_mod.all_to_all_single(recv, packed, group=group)
            dist, idx = ops.merge_packed(recv, world, 2 * pk, 0, pk, n_mine, k, metric)
        elif exchange == "allgather":
            recv = ops.empty_words(world * 2 * nq_pad * k, send)
            if isinstance(group, WorldSlice):
                raise ValueError("the all-gather exchange of a 2-D layout needs a process group (make_groups(..., exchange='allgather'))")
            dist_mod.all_gather_into_tensor(recv, send.view(-1), group=group)
            dist, idx = ops.merge_packed(recv, world, 2 * nq_pad * k, rank * pk, nq_pad * k + rank * pk, n_mine, k, metric)
        else:
            raise ValueError(f"unknown exchange '{exchange}'")
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
