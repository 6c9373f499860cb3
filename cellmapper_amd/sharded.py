"""Reference-sharded k-NN -> kernel -> transfer across the GPUs of one node.

Layout (2-D): ``world = query groups x reference shards``, rank = group * shards + shard.  A rank holds the reference
rows ``[lo, hi)`` of its shard (contiguous split), the query rows of its group (pure reference sharding = one group =
every rank holds ALL queries) and the small reference-side payloads in full (label codes, ``.obsm`` columns).  Per step,
inside a reference-shard group:

1. local exact top-k of every query against the rank's shard, with GLOBAL reference indices (``idx_offset``)
   and squared float64 distances;
2. ONE collective over RCCL/xGMI on a packed buffer (distances and indices travel together): an all-to-all by
   query slice -- rank j receives, from every rank, the local top-k of query slice j -- ``[world, 2, nq/world, k]``
   (shards x less traffic than an all-gather of the full lists; ``exchange="allgather"`` is the all-gather variant of the
   north-star wording);
3. k-way merge by (distance, global index): the same tie rule as the single-GPU search, so indices stay exact;
4. the global kernel bandwidth needs the mean (and for scarches the variance) of ALL nq*k distances: a
   3-double all-reduce over the whole world;
5. weights / normalisation on the rank's query slice -> its rows of the mapping matrix ("ELL": k columns per row).

Aggregation (second half of 8e; reference: ``cellmapper.py:338,373,595-605`` do all of it on one CPU thread):

* labels and ``.obsm`` (tens of MB): the payload is REPLICATED, every rank transfers onto its own query slice -- no
  communication (:func:`sharded_map_labels`, :func:`sharded_map_obsm`); outputs are query-sliced;
* wide ``.X`` / layers, dense or CSR: GENE COLUMNS are sharded -- rank r holds ``X[:, g_lo:g_hi]`` -- and the mapping matrix
  is replicated by ONE all-gather of its ELL rows (366 MB at 1.5 M x 30); every rank then multiplies ALL query rows with
  its slab (:func:`gather_mapping_rows` + :func:`sharded_map_layer_dense` / ``_csr``): no further communication, outputs
  are column-sliced, and a column's sum is the same sequence of multiply-adds as on one GPU (bit-identical).

The compute calls are injected (``ops``) so that the choreography can be unit-tested on CPU with gloo; in
production ``ops`` is :class:`HipOps` (device tensors through the C ABI).
"""
from __future__ import annotations

import sys

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

    def local_knn_final(self, ref_shard, qry, k, metric, idx_offset):
        """The search of a layout with ONE reference shard: final ``(dist float64 [nq, k], idx int64 [nq, k])`` -- the values
        :meth:`merge_packed` of a single part would return (euclidean: the root of the squared ranking values), without the
        packed buffer and the merge pass."""
        return self.p.knn(ref_shard, qry, k, metric=metric, idx_offset=idx_offset, f64_out=True)

    def merge_packed(self, buf, nparts, part_stride, dist_off, idx_off, nq, k, metric):
        """k-way merge of ``nparts`` lists per query living in one packed buffer (offsets / stride in 8-byte words)."""
        return self.p.merge_strided(buf, nparts, part_stride, dist_off, idx_off, nq, k, sqrt_out=(metric == "euclidean"))

    def moments(self, dist, idx, center=None):
        return self.p.moments(dist, idx, center)

    def param(self, m3, kernel):
        return self.p.kernel_param(m3, kernel)

    def weights(self, dist, idx, n_ref, kernel, param, epsilon):
        """K1-K3 on the rank's rows: ELL mapping-matrix rows ``(cols int32 [n,k], vals float32 [n,k], cnt int32 [n])``."""
        return self.p.weights(dist, idx, n_ref, kernel, param, epsilon)

    def vote(self, cols, vals, cnt, codes, n_cat):
        indptr, oc, ov = self.p.to_csr(cols, vals, cnt)
        return self.p.label_vote(indptr, oc, ov, codes, n_cat)

    def weights_vote(self, dist, idx, n_ref, kernel, param, codes, n_cat, epsilon):
        cols, vals, cnt = self.weights(dist, idx, n_ref, kernel, param, epsilon)
        return self.vote(cols, vals, cnt, codes, n_cat)

    def to_csr(self, cols, vals, cnt):
        return self.p.to_csr(cols, vals, cnt)

    def spmm_dense(self, csr, B, out=None):
        return self.p.spmm_dense(csr[0], csr[1], csr[2], B, out=out)

    def spmm_csr(self, csr, b_csr, n_out):
        """``M @ B`` for a CSR payload (two-pass form: exact allocation) -> ``(out_indptr, out_cols, out_vals)``."""
        return self.p.spmm_csr(csr[0], csr[1], csr[2], b_csr[0], b_csr[1], b_csr[2], n_out)

    def empty_words(self, n, like):
        return self.t.empty((n,), dtype=self.t.int64, device=like.device)

    def empty_i32(self, n, like):
        return self.t.empty((n,), dtype=self.t.int32, device=like.device)

    def synchronize(self):
        self.t.cuda.synchronize()

    def cat(self, parts):
        return self.t.cat(parts)

    def equal(self, a, b):
        return bool(self.t.equal(a, b))

    def flag(self, ok: bool, like):
        return self.t.tensor([1 if ok else 0], dtype=self.t.int32, device=like.device)


REF_SHARD_BUDGET_BYTES = 32 << 30   # a reference shard larger than this is split further (MI355X: 288 GB; the search's tile image, the
                                    # per-query workspace and the payloads want the rest)


def grid_2d(world: int, ref_shards: int | None = None, ref_bytes: int | None = None, budget: int = REF_SHARD_BUDGET_BYTES):
    """(query groups, reference shards) with ``groups * shards == world``.

    Default (round 5, from measurements -- profiles/r05_rank_shares.md): the FEWEST reference shards whose shard fits
    ``budget`` bytes of a GPU, all remaining parallelism on the QUERY axis.  One rank's share of BASELINE config 2 on the
    current library, per query searched: 0.030 us per shard visited + 8.5e-8 us per reference row (d = 50; d = 128: 0.029 us +
    2.1e-7) -- the filter's candidate handling (appends, merges, re-rank) is paid once per (query, shard) and does not shrink
    with the shard, so splitting the reference R ways adds R x that overhead in total while splitting the queries adds
    nothing and needs no exchange.  At 8 GPUs: 8 query groups x 1 shard 37 ms per step, 4 x 2 45 ms, 2 x 4 48 ms, pure
    reference sharding (1 x 8) 65 ms.  A reference that does not fit one GPU (``ref_bytes / shards > budget``) is split
    as far as needed; ``ref_shards=world`` forces the pure reference sharding of the north-star wording (BASELINE config 4
    is DEFINED that way and bench.py runs it so), any other value the 2-D layout.  Without ``ref_bytes`` the reference is
    taken to fit.
    """
    if ref_shards is None:
        ref_shards = 1
        if ref_bytes is not None:
            while ref_shards < world and (world % ref_shards or ref_bytes / ref_shards > budget):
                ref_shards += 1
    ref_shards = max(1, min(world, int(ref_shards)))
    while world % ref_shards:
        ref_shards -= 1
    return world // ref_shards, ref_shards


class Layout:
    """Who owns what in the 2-D layout: ``rank = group * ref_shards + ref_rank``.  All row numbers are GLOBAL query rows."""

    def __init__(self, world: int, rank: int, nq_total: int, ref_shards: int | None = None, ref_bytes: int | None = None):
        self.world, self.rank, self.nq_total = int(world), int(rank), int(nq_total)
        self.n_groups, self.ref_shards = grid_2d(world, ref_shards, ref_bytes)
        self.group, self.ref_rank = rank // self.ref_shards, rank % self.ref_shards
        self.group_lo, self.group_hi = shard_bounds(nq_total, self.n_groups, self.group)   # the queries this rank searches

    def slice_of(self, r: int):
        """Global query rows [lo, hi) whose merged lists (and mapping-matrix rows) rank ``r`` owns after the exchange."""
        g, s = r // self.ref_shards, r % self.ref_shards
        g_lo, g_hi = shard_bounds(self.nq_total, self.n_groups, g)
        lo, hi = slice_bounds(g_hi - g_lo, self.ref_shards, s)
        return g_lo + lo, g_lo + hi

    def all_slices(self):
        return [self.slice_of(r) for r in range(self.world)]

    @property
    def my_slice(self):
        return self.slice_of(self.rank)


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


SUPPORTED_METRICS = ("euclidean", "sqeuclidean", "cosine", "correlation", "manhattan", "chebyshev", "canberra", "braycurtis",
                     "hamming", "jaccard", "dice", "rogerstanimoto", "russellrao", "sokalmichener", "sokalsneath", "yule", "haversine",
                     "nan_euclidean")


class ExchangeState:
    """Which exchange a run uses, decided on its FIRST step.  The default all-to-all of the 2-D layout (split sizes that are
    zero outside the rank's reference-shard group) could never be exercised between two real GPUs before the first hardware
    run, so that step runs it NEXT TO the plain world-wide all-gather -- every rank receives every rank's lists through the
    default communicator, the least exotic collective there is -- and compares the two merged results bit for bit.  If the
    all-to-all raises, or the results differ on any rank (one all-reduce of a flag), the run continues on the all-gather
    in the same process group: nothing is re-executed, and the bench line says which exchange ran and why."""

    def __init__(self, requested: str = "alltoall", selfcheck: bool = True):
        self.requested = requested
        self.mode = requested            # "alltoall" | "allgather" | "allgather_world"
        self.selfcheck = selfcheck and requested == "alltoall"
        self.checked = False
        self.note = None                 # why the run left the requested exchange (None = it did not)
        self.pad_words = None            # world all-gather: words per rank (max over the query groups)
        self.timer = None                # optional: an object with .mark(name) (bench.py: stream events around the exchange)
        self.sent_words = 0              # 8-byte words this rank handed to the last exchange collective (own chunk included)

    def mark(self, name: str):
        if self.timer is not None:
            self.timer.mark(name)

    def describe(self):
        return {"requested": self.requested, "used": self.mode, "self_check": ("passed" if self.checked and self.note is None
                else ("not run" if not self.checked else "FAILED: " + str(self.note)))}


def _world_allgather_merge(ops, dist_mod, send, group, world, rank, nq_pad, k, n_mine, metric, state):
    """The exchange through ONE all-gather of the whole world (default communicator): rank r's packed buffer lands at
    ``r * pad`` words; the lists of this rank's query slice are merged out of its group's parts."""
    n_all = dist_mod.get_world_size()
    words = 2 * nq_pad * k
    if state.pad_words is None:   # groups of a ragged split differ by one query: pad every rank's buffer to the largest
        w = ops.flag(True, send).to(dtype=send.dtype) * words
        dist_mod.all_reduce(w, op=dist_mod.ReduceOp.MAX)
        state.pad_words = int(w.item())
    pad = state.pad_words
    flat = send.view(-1)
    if pad != words:
        padded = ops.empty_words(pad, send)
        padded[:words] = flat
        flat = padded
    recv = ops.empty_words(n_all * pad, send)
    dist_mod.all_gather_into_tensor(recv, flat)
    first = group.first if isinstance(group, WorldSlice) else 0
    pk = (nq_pad // world) * k
    return ops.merge_packed(recv, world, pad, first * pad + rank * pk, first * pad + nq_pad * k + rank * pk, n_mine, k, metric)


def sharded_knn(ops, dist_mod, ref_shard, shard_lo, qry, k, metric="euclidean", exchange="alltoall", group=None,
                ref_world=None, ref_rank=None, state: ExchangeState | None = None):
    """Steps 1-3 of the module docstring.  Returns ``(dist, idx, (q_lo, q_hi))``: the merged exact lists of this rank's query
    slice (``(q_lo, q_hi)`` relative to ``qry``).  ``exchange`` may be an :class:`ExchangeState` (first-step self-check and
    fall-back, see there) passed as ``state``; a plain string runs that exchange unchecked."""
    if metric not in SUPPORTED_METRICS:
        raise ValueError(f"Metric '{metric}' is not supported by the sharded hip backend (supported: {', '.join(SUPPORTED_METRICS)}).")
    n_all = dist_mod.get_world_size()
    world = n_all if ref_world is None else ref_world
    rank = dist_mod.get_rank() if ref_rank is None else ref_rank
    mode = state.mode if state is not None else exchange
    nq = qry.shape[0]
    per = (nq + world - 1) // world
    nq_pad = per * world
    if state is not None:
        state.mark("search")
    if world == 1 and hasattr(ops, "local_knn_final"):
        # one reference shard (the default layout when the reference fits a GPU): nothing to exchange or merge -- the search
        # writes the final lists itself (float64 values, root taken for euclidean: what the merge of one part would produce)
        dist, idx = ops.local_knn_final(ref_shard, qry, k, metric, shard_lo)
        if state is not None:
            state.mark("exchange")
            state.sent_words = 0
            state.mark("merged")
        return dist, idx, (0, nq)
    send = ops.local_knn_packed(ref_shard, qry, k, metric, shard_lo, nq_pad)   # [2, nq_pad, k] 8-byte words
    if state is not None:
        state.mark("exchange")
        state.sent_words = 2 * nq_pad * k * (world if (state.mode != "alltoall") else 1)
    pk = per * k
    if world == 1:
        dist, idx = ops.merge_packed(send, 1, 2 * nq_pad * k, 0, nq_pad * k, nq, k, metric)
        if state is not None:
            state.mark("merged")
        return dist, idx, (0, nq)
    q_lo, q_hi = slice_bounds(nq, world, rank)
    n_mine = max(q_hi - q_lo, 0)

    def run_alltoall():
        # chunk j of the send buffer = what rank j merges: [2][per][k] of query slice j
        packed = send.view(2, world, pk).transpose(0, 1).contiguous().view(-1)
        recv = ops.empty_words(world * 2 * pk, send)
        if isinstance(group, WorldSlice):   # the group's all-to-all inside the world's: empty chunks for everybody else
            splits = [2 * pk if group.first <= r < group.first + group.count else 0 for r in range(n_all)]
            dist_mod.all_to_all_single(recv, packed, output_split_sizes=splits, input_split_sizes=splits)
        else:
            dist_mod.all_to_all_single(recv, packed, group=group)
        return ops.merge_packed(recv, world, 2 * pk, 0, pk, n_mine, k, metric)

    def run_allgather():
        if isinstance(group, WorldSlice):
            raise ValueError("the all-gather exchange of a 2-D layout needs a process group (make_groups(..., exchange='allgather'))")
        recv = ops.empty_words(world * 2 * nq_pad * k, send)
        dist_mod.all_gather_into_tensor(recv, send.view(-1), group=group)
        return ops.merge_packed(recv, world, 2 * nq_pad * k, rank * pk, nq_pad * k + rank * pk, n_mine, k, metric)

    if mode == "alltoall":
        if state is not None and state.selfcheck and not state.checked:
            ok, err, got = True, None, None
            # What this catches: an exchange that RETURNS wrong lists, and a call the backend REJECTS (argument validation
            # raises synchronously and identically on every rank: the arguments are symmetric -- that is what gloo's
            # "not implemented" and RCCL's invalid-argument checks do).  It does not make an asynchronous RCCL failure
            # survivable: that surfaces at the synchronisation below (or in the watchdog) with the communicator already
            # unusable, so the run then ends with the error instead of limping on -- see the timeout in bench.py.
            try:
                got = run_alltoall()
                sync = getattr(ops, "synchronize", None)
                if sync is not None:
                    sync()   # surface asynchronous errors of the collective HERE, inside the try
            except Exception as e:
                ok, err = False, f"{type(e).__name__}: {e}"
                if "NCCL" in str(e) or "RCCL" in str(e) or "ncclInternalError" in str(e) or "unhandled" in str(e).lower():
                    raise   # a failed communicator cannot carry the fall-back either
            want = _world_allgather_merge(ops, dist_mod, send, group, world, rank, nq_pad, k, n_mine, metric, state)
            if ok:
                ok = ops.equal(got[0], want[0]) and ops.equal(got[1], want[1])
                if not ok:
                    err = "merged lists differ from the world all-gather route"
            f = ops.flag(ok, send)
            dist_mod.all_reduce(f, op=dist_mod.ReduceOp.MIN)
            state.checked = True
            if int(f.item()) == 0:
                state.mode, state.note = "allgather_world", err or "another rank reported a mismatch"
                print(f"[sharded] rank {dist_mod.get_rank()}: all-to-all exchange failed its self-check ({state.note}); "
                      f"continuing on the world all-gather", file=sys.stderr, flush=True)
            dist, idx = want
        else:
            dist, idx = run_alltoall()
    elif mode == "allgather":
        dist, idx = run_allgather()
    elif mode == "allgather_world":
        st = state if state is not None else ExchangeState("allgather_world", selfcheck=False)
        dist, idx = _world_allgather_merge(ops, dist_mod, send, group, world, rank, nq_pad, k, n_mine, metric, st)
    else:
        raise ValueError(f"unknown exchange '{mode}'")
    if state is not None:
        state.mark("merged")
    return dist, idx, (q_lo, q_hi)


def global_kernel_param(ops, dist_mod, dist, idx, kernel):
    """Step 4: the reference's GLOBAL bandwidth (``knn.py:194-209``) from moments all-reduced over the whole world."""
    if kernel not in ("gaussian", "scarches"):
        return None
    n_all = dist_mod.get_world_size()
    m = ops.moments(dist, idx)
    if n_all > 1:
        dist_mod.all_reduce(m)
    param = ops.param(m, "gaussian")  # mean distance
    if kernel == "scarches":
        m2 = ops.moments(dist, idx, center=param)
        if n_all > 1:
            dist_mod.all_reduce(m2)
        param = ops.param(m2, "scarches")
    return param


def sharded_mapping_rows(ops, dist_mod, ref_shard, shard_lo, n_ref_total, qry, k, metric="euclidean", kernel="gaussian",
                         epsilon=1e-8, exchange="alltoall", group=None, ref_world=None, ref_rank=None, state=None):
    """Steps 1-5: this rank's rows of the row-normalised mapping matrix.  Returns a dict: ``cols / vals / cnt`` (ELL),
    ``dist / idx`` (merged lists), ``bounds`` (slice relative to ``qry``), ``param``."""
    dist, idx, qb = sharded_knn(ops, dist_mod, ref_shard, shard_lo, qry, k, metric, exchange, group, ref_world, ref_rank, state)
    param = global_kernel_param(ops, dist_mod, dist, idx, kernel)
    cols, vals, cnt = ops.weights(dist, idx, n_ref_total, kernel, param, epsilon)
    return dict(cols=cols, vals=vals, cnt=cnt, dist=dist, idx=idx, bounds=qb, param=param)


def sharded_map_labels(ops, dist_mod, ref_shard, shard_lo, n_ref_total, qry, k, codes, n_cat, metric="euclidean",
                       kernel="gaussian", epsilon=1e-8, exchange="alltoall", group=None, ref_world=None, ref_rank=None,
                       state=None):
    """Run one sharded step.  Returns ``(code, conf, (q_lo, q_hi), dist, idx)`` for this rank's query slice.

    ``dist_mod`` is ``torch.distributed`` (already initialised); ``ops`` supplies compute (HipOps in production).
    ``qry`` holds the queries of this rank's QUERY GROUP; ``group`` / ``ref_world`` / ``ref_rank`` describe the
    reference-shard group the exchange runs in (default: the whole world = pure reference sharding).  The bandwidth
    statistics are always reduced over the whole world.  ``(q_lo, q_hi)`` is relative to ``qry``.

    The exchange is ONE collective on a packed buffer (ranking values and global indices travel together as 8-byte
    words): ``all_to_all_single`` of ``[world][2][per][k]`` (default), or ``all_gather_into_tensor`` of ``[2][nq_pad][k]``.
    Shard-local lists carry the float64 values the single-GPU search ranks by (squared euclidean; float64 cosine even
    for float32 input), so the merged order -- and therefore every index -- equals the unsharded result.
    Label codes are replicated (A0, ``cellmapper.py:589-623``): the vote needs no communication.
    """
    if metric not in SUPPORTED_METRICS:
        raise ValueError(f"Metric '{metric}' is not supported by the sharded hip backend (supported: {', '.join(SUPPORTED_METRICS)}).")
    r = sharded_mapping_rows(ops, dist_mod, ref_shard, shard_lo, n_ref_total, qry, k, metric, kernel, epsilon, exchange, group,
                             ref_world, ref_rank, state)
    code, conf = ops.vote(r["cols"], r["vals"], r["cnt"], codes, n_cat)
    return code, conf, r["bounds"], r["dist"], r["idx"]


def sharded_map_obsm(ops, rows, payload):
    """A1 / A0n on this rank's query slice with a REPLICATED dense payload ``[n_ref, d_out]`` (``cellmapper.py:338,628``):
    no communication; the output rows are those of ``rows['bounds']``."""
    return ops.spmm_dense(ops.to_csr(rows["cols"], rows["vals"], rows["cnt"]), payload)


def gather_mapping_rows(ops, dist_mod, rows, layout: Layout):
    """Replicate the mapping matrix: ONE all-gather (whole world, default communicator) of every rank's ELL rows, packed as
    int32 words ``[per_max][k] columns | [per_max][k] value bits | [per_max] counts``.  Returns ``(cols [nq, k], vals [nq, k],
    cnt [nq])`` in GLOBAL query order (8 bytes per entry: 366 MB at 1.5 M x 30)."""
    cols, vals, cnt = rows["cols"], rows["vals"], rows["cnt"]
    k = cols.shape[1]
    if layout.world == 1:
        return cols, vals, cnt
    slices = layout.all_slices()
    per_max = max(hi - lo for lo, hi in slices)
    n_mine = cols.shape[0]
    if n_mine != slices[layout.rank][1] - slices[layout.rank][0]:
        raise ValueError("gather_mapping_rows: this rank's rows do not match its slice of the layout")
    words = per_max * (2 * k + 1)
    send = ops.empty_i32(words, cols)
    if n_mine < per_max:
        send.zero_()
    send[:per_max * k].view(per_max, k)[:n_mine] = cols
    send[per_max * k:2 * per_max * k].view(per_max, k)[:n_mine] = vals.view(cols.dtype)
    send[2 * per_max * k:][:n_mine] = cnt
    recv = ops.empty_i32(layout.world * words, cols)
    dist_mod.all_gather_into_tensor(recv, send)
    parts_c, parts_v, parts_n = [], [], []
    for r, (lo, hi) in enumerate(slices):
        base = r * words
        n = hi - lo
        parts_c.append(recv[base:base + per_max * k].view(per_max, k)[:n])
        parts_v.append(recv[base + per_max * k:base + 2 * per_max * k].view(per_max, k)[:n])
        parts_n.append(recv[base + 2 * per_max * k:base + words][:n])
    return ops.cat(parts_c), ops.cat(parts_v).view(vals.dtype), ops.cat(parts_n)


def sharded_map_layer_dense(ops, full_rows, payload_slab, out=None):
    """A2 with gene-column sharding (``cellmapper.py:372-373``): ``M @ X[:, g_lo:g_hi]`` for ALL query rows; ``full_rows`` is
    :func:`gather_mapping_rows`' result (or its CSR form), ``payload_slab`` the rank's columns ``[n_ref, g_hi - g_lo]``."""
    csr = full_rows if len(full_rows) == 3 and full_rows[0].dim() == 1 else ops.to_csr(*full_rows)
    return ops.spmm_dense(csr, payload_slab, out=out)


def sharded_map_layer_csr(ops, full_rows, payload_slab_csr, n_out_slab):
    """A3 with gene-column sharding (``cellmapper.py:372-373``, scipy ``csr_matmat``): ``M @ X[:, g_lo:g_hi]`` for a CSR payload
    slab ``(indptr, cols - g_lo, vals)`` with ``n_out_slab`` columns -> the slab of the product as CSR (columns relative to the
    slab).  Exact zeros are dropped per output entry, so the slab equals the same columns of the one-GPU product."""
    csr = full_rows if len(full_rows) == 3 and full_rows[0].dim() == 1 else ops.to_csr(*full_rows)
    return ops.spmm_csr(csr, payload_slab_csr, n_out_slab)
