"""bench.py --config 3 / 5: the aggregation-heavy workloads, on one GPU or -- gene-column sharded -- on N.

config 3: 1.5 M x 1.5 M, d=50, k=30 search, then obsm transfer (d_out = 2) and the dense layer transfer over reference.X
          with 20 000 genes (A1 + A2).  X (120 GB float32) is resident in HBM; the product (another 120 GB) is produced in
          query tiles -- resident as a whole when it fits next to X, otherwise through a two-tile ring that a consumer
          (D2H copy, on-device evaluation) would drain.
config 5: self-mapping of 4 M spatial cells (d=2, k=90; uniform-grid search) and expression imputation from a sparse
          reference.X (30 000 genes, 3 % density, CSR): `M @ X` as CSR (scipy csr_matmat semantics), produced in query tiles
          of fixed capacity -- the whole product (~0.9 TB) is never resident.

N > 1 (second half): the search runs reference-sharded exactly as in config 2 (cellmapper_amd/sharded.py); every
rank then owns a slice of the mapping-matrix rows.  `.obsm` is replicated and transferred per query slice (no
communication); for the wide layer the mapping matrix is replicated by ONE all-gather of its ELL rows and every rank
multiplies ALL query rows with ITS gene columns (`X[:, g_lo:g_hi]`, generated in place: the streams are counter-based) -- no
further communication.  A column's sum is the same sequence of multiply-adds whatever the slab, so the N > 1 product equals
the N = 1 product bit for bit: `layer_checksum` (a position-weighted sum of the output bits over all ranks) is the same
number at every N.

The roofline object describes the dominant kernel of each workload, an HBM-bound gather: ALGORITHMIC bytes per query row
(A2 `k*8 + k*G*4 + G*4`; A3 `k*8 + sum(neighbour nnz)*8 + out_nnz*8`, G = the rank's gene columns) x rows /
kernel time (HIP events on the library's stream), against 8 TB/s.
"""
from __future__ import annotations

import os
import time

import numpy as np

import bench_data as bd

ROOT = os.path.dirname(os.path.abspath(__file__))
PEAK_HBM_GBS = 8000.0
CHK_MOD = 1000003


def run(ctx):
    return run_config3(ctx) if ctx.a.config == 3 else run_config5(ctx)


def _mapping_rows(ctx, ref, qry, metric="euclidean"):
    """This rank's rows of the mapping matrix (ELL) + merged lists: the sharded choreography at N > 1, the same calls on one
    GPU otherwise."""
    from cellmapper_amd.sharded import sharded_mapping_rows

    a, lay = ctx.a, ctx.lay
    if ctx.use_dist:
        return sharded_mapping_rows(ctx.ops, ctx.dist, ref, ctx.ref_lo, a.nref, qry, a.k, metric=metric, kernel=a.kernel,
                                    exchange=a.exchange, group=ctx.pg, ref_world=lay.ref_shards, ref_rank=lay.ref_rank, state=ctx.state)
    pipe = ctx.pipe
    dist, idx = pipe.knn(ref, qry, a.k)
    m = pipe.moments(dist, idx)
    param = pipe.kernel_param(m, a.kernel) if a.kernel in ("gaussian", "scarches") else None
    cols, vals, cnt = pipe.weights(dist, idx, a.nref, a.kernel, param)
    return dict(cols=cols, vals=vals, cnt=cnt, dist=dist, idx=idx, bounds=(0, qry.shape[0]), param=param)


def _full_csr(ctx, rows):
    from cellmapper_amd.sharded import gather_mapping_rows

    full = gather_mapping_rows(ctx.ops, ctx.dist, rows, ctx.lay) if ctx.use_dist else (rows["cols"], rows["vals"], rows["cnt"])
    return ctx.ops.to_csr(*full)


def _knn_gate(ctx, rows, ref_host_fn, qry_rows_fn, what):
    """k-NN part of the parity gate (+ the timed oracle run = the search part of `cpu_baseline`)."""
    import bench

    a = ctx.a
    if a.check <= 0 and a.cpu_sample <= 0:
        return None, None, True
    my_lo, my_hi = ctx.lay.my_slice if ctx.use_dist else (0, a.nq)
    return bench.parity_gate(ctx, rows["dist"], rows["idx"], None, None, my_lo, my_hi, a.cpu_sample > 0, ref_host_fn, qry_rows_fn, what)


def _combine_cpu(cb_knn, rec, agg_s_per_row, agg_what, threads):
    """End-to-end CPU rate of search + aggregation from the two bounded legs (both linear in the number of query rows)."""
    if cb_knn is None or rec is None:
        return None
    s_knn = rec["seconds_per_query_cpu"]
    return dict(value=1.0 / (s_knn + agg_s_per_row), unit="query cells/s", cores=threads, kind="port",
                sample=cb_knn["sample"] + f"; + {agg_what}: {agg_s_per_row * 1e3:.3f} ms per query row; both legs are linear in the "
                                          f"number of query rows, the rate is 1 / (their per-row times)")


# --------------------------------------------------------------------------------------------- config 3 ----------
def run_config3(ctx):
    a, torch, pipe, dev, lay = ctx.a, ctx.torch, ctx.pipe, ctx.dev, ctx.lay
    from cellmapper_amd.sharded import shard_bounds, sharded_map_obsm

    G = a.genes
    g_lo, g_hi = shard_bounds(G, ctx.world, ctx.rank)          # this rank's gene columns
    gw = g_hi - g_lo
    ref = bd.fill_rows(torch, "normal", bd.STREAM_REF, ctx.ref_lo, ctx.ref_hi, a.d, dev)
    qry = bd.fill_rows(torch, "normal", bd.STREAM_QRY, lay.group_lo, lay.group_hi, a.d, dev)
    umap = bd.fill_rows(torch, "normal", bd.STREAM_OBSM, 0, a.nref, 2, dev)           # replicated
    X = bd.fill_rows(torch, "uniform", bd.STREAM_X, 0, a.nref, gw, dev, c0=g_lo)      # reference.X[:, g_lo:g_hi], dense float32
    torch.cuda.synchronize()
    free_b, _ = torch.cuda.mem_get_info(dev)
    out_bytes = a.nq * gw * 4
    resident = out_bytes + (6 << 30) < free_b     # the whole slab of the product stays in HBM (plus the search's workspace)
    tile = a.tile or (a.nq if resident else max(1024, min(a.nq, (free_b // 4) // (gw * 4 * 2))))
    out = torch.empty((a.nq if resident else 2 * tile, gw), dtype=torch.float32, device=dev)
    torch.cuda.set_stream(pipe.stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    spmm_ms = []

    def step():
        rows = _mapping_rows(ctx, ref, qry)
        obsm = sharded_map_obsm(ctx.ops, rows, umap)                      # A1 on the rank's query slice
        indptr, oc, ov = _full_csr(ctx, rows)                             # ONE all-gather at N > 1
        e0.record(pipe.stream)
        for i, q0 in enumerate(range(0, a.nq, tile)):                     # A2, query tiles x the rank's gene columns
            q1 = min(a.nq, q0 + tile)
            dst = out[q0:q1] if resident else out[(i & 1) * tile:(i & 1) * tile + (q1 - q0)]
            pipe.spmm_dense(indptr[q0:q1 + 1], oc, ov, X, out=dst)
        e1.record(pipe.stream)
        return rows, (indptr, oc, ov), obsm

    tf0 = time.perf_counter()
    res = step()
    pipe.synchronize()
    first_call_ms = (time.perf_counter() - tf0) * 1e3
    for _ in range(max(0, a.warmup - 1)):
        res = step()
    pipe.synchronize()
    if ctx.use_dist:
        ctx.dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
        pipe.synchronize()
        spmm_ms.append(e0.elapsed_time(e1))
    if ctx.use_dist:
        ctx.dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if ctx.world > 1:
        ctx.dist.all_reduce(tmax, op=ctx.dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    stages = pipe.h.stage_times()
    ms = float(np.mean(spmm_ms))
    bytes_per_q = a.k * 8 + a.k * gw * 4 + gw * 4
    achieved = bytes_per_q * a.nq / (ms * 1e-3) / 1e9
    rows, (indptr, oc, ov), obsm = res
    # position-weighted checksum of the layer product over ALL ranks' slabs (sampled rows x every gene): equal at every N
    ns_c = min(4096, a.nq)
    srows = torch.from_numpy((np.arange(ns_c, dtype=np.int64) * a.nq) // ns_c).to(dev)
    if resident:
        blk = out[srows]
    else:
        blk = torch.cat([pipe.spmm_dense(indptr[i:i + 2], oc, ov, X) for i in srows.tolist()[:256]])
        srows = srows[:256]
    wts = ((srows[:, None] * G + torch.arange(g_lo, g_hi, device=dev, dtype=torch.int64)[None, :]) % CHK_MOD) + 1
    chk = (blk.view(torch.int32).to(torch.int64) * wts).sum().reshape(1)
    if ctx.world > 1:
        ctx.dist.all_reduce(chk)
    result = None
    if ctx.rank == 0:
        result = {
            "metric": "query cells/s end-to-end (k-NN+kernel+transfer)", "value": a.nq * a.steps / elapsed, "unit": "query cells/s",
            "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "timed_region": "device-resident",
            "first_call_ms": first_call_ms,
            "config": {"workload": f"{a.nq} query x {a.nref} reference cells, d={a.d}, k={a.k}, {a.kernel} kernel, obsm (d_out=2) + dense "
                                   f"layer X ({G} genes, float32) transfer (BASELINE configs[2]); {ctx.world} GPU(s): search "
                                   f"{ctx.parallelism()}, layer product gene-column sharded ({gw} genes on rank 0); output "
                                   f"{'resident in HBM' if resident else f'in a ring of two {tile}-row tiles'}",
                       "baseline_config": 3, "parallelism": ctx.parallelism() + f"+genes/{ctx.world}",
                       "exchange": ctx.state.describe() if (ctx.use_dist and lay.ref_shards > 1) else None,
                       "nq": a.nq, "n_ref": a.nref, "d": a.d, "k": a.k, "genes": G, "genes_per_rank": gw, "query_tile": tile,
                       "output_resident": bool(resident),
                       "data_generator": "counter-based hash streams (bench_data.py), N-independent"},
            "roofline": {"bound": "hbm", "kernel": "spmm_wide<float,4> (k-sparse x dense gather-accumulate, A2)", "achieved": achieved,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS, "traffic": None, "kernel_ms": ms,
                         "bytes_per_query": bytes_per_q,
                         "kernel_ms_covers": "all query tiles of the dense layer product in one step (rank 0's gene columns), mean over the timed steps"},
            "stage_ms_last_step": dict(stages, layer_spmm=ms),
            "layer_checksum": {"value": f"{int(chk.item()) & 0xFFFFFFFFFFFFFFFF:016x}", "rows": int(srows.shape[0]),
                               "what": "sum over sampled rows x all genes of float32 bits x ((row * G + gene) mod 1000003 + 1), all ranks; "
                                       "identical at every N"},
        }
    ok = True
    # parity: (1) merged k-NN lists of every rank's slice against the CPU oracle (+ the timed search leg); (2) layer / obsm rows
    # against the oracle's csr_matvecs restatement on every rank's slab
    rec, cb_knn, ok_knn = _knn_gate(ctx, rows, None, None, "k-NN+kernel")
    ok = ok and ok_knn
    if a.check > 0:
        from oracle import oracle as orc

        orc.build()
        threads = orc.set_num_threads(__import__("bench").usable_cores())
        ns = min(128, a.nq)
        sel_np = (np.arange(ns, dtype=np.int64) * a.nq) // ns
        ipt = indptr.cpu().numpy()
        rows_c = np.concatenate([np.arange(ipt[i], ipt[i + 1]) for i in sel_np])
        c_np = oc[torch.from_numpy(rows_c).to(dev)].cpu().numpy()
        v_np = ov[torch.from_numpy(rows_c).to(dev)].cpu().numpy()
        uniq, inv = np.unique(c_np, return_inverse=True)
        Xs = X[torch.from_numpy(uniq).to(dev).long()].cpu().numpy()      # only the gathered reference rows travel
        lens = np.array([ipt[i + 1] - ipt[i] for i in sel_np])
        ip2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        # (the oracle accumulates in stored order; remapped columns keep the order of the original ascending columns)
        tq0 = time.perf_counter()
        want = orc.spmm_dense(ip2, inv.astype(np.int32), v_np, Xs)
        agg_s = (time.perf_counter() - tq0) / ns * (G / max(gw, 1))      # per query row, scaled to all G genes
        sel = torch.from_numpy(sel_np).to(dev)
        if resident:
            got = out[sel].cpu().numpy()
        else:
            got = torch.cat([pipe.spmm_dense(indptr[i:i + 2], oc, ov, X) for i in sel_np.tolist()]).cpu().numpy()
        ok_layer = bool(np.array_equal(got, want))
        # obsm: the rank's own query slice
        q_lo, q_hi = (lay.my_slice if ctx.use_dist else (0, a.nq))
        mine = sel_np[(sel_np >= q_lo) & (sel_np < q_hi)]
        ok_obsm = True
        if len(mine):
            um = umap.cpu().numpy()
            for i in mine.tolist():
                r = slice(ipt[i], ipt[i + 1])
                w = orc.spmm_dense(np.array([0, ipt[i + 1] - ipt[i]], np.int64), oc[r].cpu().numpy(), ov[r].cpu().numpy(), um)
                ok_obsm = ok_obsm and bool(np.array_equal(obsm[i - q_lo].cpu().numpy(), w[0]))
        ok_all = ctx.all_true(ok_layer and ok_obsm)
        ok = ok and ok_all
        if ctx.rank == 0:
            result["parity_check"] = {"knn": rec, "layer_rows_checked_per_rank": int(ns), "layer_rows_bit_exact": ok_layer,
                                      "obsm_rows_bit_exact": ok_obsm, "all_ranks_ok": ok_all, "ok": bool(ok)}
            cb = _combine_cpu(cb_knn, rec, agg_s, f"oracle csr_matvecs restatement on {ns} rows x {gw} genes (scaled to {G} genes), "
                                                  f"{threads} OpenMP threads", threads)
            if cb is not None:
                result["cpu_baseline"] = cb
    return result, ok


# --------------------------------------------------------------------------------------------- config 5 ----------
def run_config5(ctx):
    a, torch, pipe, dev, lay = ctx.a, ctx.torch, ctx.pipe, ctx.dev, ctx.lay
    from cellmapper_amd.sharded import shard_bounds

    G = a.genes
    n = a.nq                                  # self-mapping: query = reference
    a.nref = n
    per_all = max(1, int(round(a.density * G)))
    width = G // per_all
    s_lo, s_hi = shard_bounds(per_all, ctx.world, ctx.rank)       # this rank's strata of genes = columns [s_lo*width, s_hi*width)
    ref_lo, ref_hi = shard_bounds(n, lay.ref_shards, lay.ref_rank)
    xy_ref = bd.fill_rows(torch, "uniform", bd.STREAM_XY, ref_lo, ref_hi, a.d, dev, scale=2000.0)
    xy_qry = bd.fill_rows(torch, "uniform", bd.STREAM_XY, lay.group_lo, lay.group_hi, a.d, dev, scale=2000.0)
    b_indptr, b_cols, b_vals, per, g0 = bd.sparse_rows(torch, 0, n, G, a.density, dev, s0=s_lo, s1=s_hi)
    Gs = (s_hi - s_lo) * width + (G - per_all * width if s_hi == per_all else 0)   # columns of the slab (the tail genes are empty)
    torch.cuda.synchronize()
    # output tile: worst case min(Gs, k * per) entries per row; two tiles alternate (the consumer drains one while the next
    # is produced)
    row_cap = min(Gs, a.k * per)
    free_b, _ = torch.cuda.mem_get_info(dev)
    tile = a.tile or max(256, min(n, 16384, (free_b // 4) // (row_cap * 8 * 2)))
    t_cols = [torch.empty(tile * row_cap, dtype=torch.int32, device=dev) for _ in range(2)]
    t_vals = [torch.empty(tile * row_cap, dtype=torch.float32, device=dev) for _ in range(2)]
    t_ptr = [torch.empty(tile + 1, dtype=torch.int64, device=dev) for _ in range(2)]
    t_rows = [torch.empty(tile, dtype=torch.int32, device=dev) for _ in range(2)]
    out_nnz = torch.zeros(1, dtype=torch.int64, device=dev)
    chk = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.set_stream(pipe.stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a3_ms = []

    use_order = not os.environ.get("CELLMAP_NO_ROW_ORDER")
    one_pass = Gs <= pipe.SINGLE_PASS_MAX_COLS and not os.environ.get("CELLMAP_A3_TWO_PASS") and not os.environ.get("CELLMAP_SPGEMM_CT")
    # CELLMAP_A3_COMPLETION: rows of a tile in completion order (out_rows names them; ~6 % faster) instead of the order's own
    # sequence (offsets from the chained scan)
    any_order = one_pass and bool(os.environ.get("CELLMAP_A3_COMPLETION"))
    sliced = one_pass and not os.environ.get("CELLMAP_A3_BARRIER")   # wave-sliced kernel (payload index) unless asked otherwise
    last = {}
    old_ref_bounds = (ctx.ref_lo, ctx.ref_hi)
    ctx.ref_lo, ctx.ref_hi = ref_lo, ref_hi

    def step(checksum=False):
        rows = _mapping_rows(ctx, xy_ref, xy_qry)
        indptr, oc, ov = _full_csr(ctx, rows)
        out_nnz.zero_()
        e0.record(pipe.stream)
        # rows in LOCALITY order (inside the timed region): a tile's rows share neighbours, the gathered payload rows stay
        # in L2; tile i holds the product rows of mapping-matrix rows order[q0:q1]
        order = pipe.locality_order(indptr, oc, n) if use_order else None
        # payload index (a property of reference.X, rebuilt every step so that it is inside the timed region)
        pidx = pipe.payload_index(b_indptr, b_cols, Gs) if sliced else None
        for i, q0 in enumerate(range(0, n, tile)):
            q1 = min(n, q0 + tile)
            b = i & 1
            pipe.spmm_csr(indptr, oc, ov, b_indptr, b_cols, b_vals, Gs, q0, q1, out=(t_ptr[b], t_cols[b], t_vals[b]), order=order,
                          out_rows=t_rows[b] if any_order else None, payload_index=pidx)
            out_nnz.add_(t_ptr[b][q1 - q0])
            if checksum:   # untimed pass: every entry's float bits x a weight of its GLOBAL gene column (order-independent sum)
                nz = int(t_ptr[b][q1 - q0].item())
                chk.add_((t_vals[b][:nz].view(torch.int32).to(torch.int64) * (((t_cols[b][:nz].to(torch.int64) + g0) % CHK_MOD) + 1)).sum())
            last.update(q0=q0, q1=q1, b=b, order=order)
        e1.record(pipe.stream)
        return rows, (indptr, oc, ov)

    tf0 = time.perf_counter()
    res = step()
    pipe.synchronize()
    first_call_ms = (time.perf_counter() - tf0) * 1e3
    for _ in range(max(0, a.warmup - 1)):
        res = step()
    pipe.synchronize()
    if ctx.use_dist:
        ctx.dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
        pipe.synchronize()
        a3_ms.append(e0.elapsed_time(e1))
    if ctx.use_dist:
        ctx.dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if ctx.world > 1:
        ctx.dist.all_reduce(tmax, op=ctx.dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    stages = pipe.h.stage_times()
    ms = float(np.mean(a3_ms))
    nnz_rank = int(out_nnz.item())
    res = step(checksum=True)          # untimed: the checksum of every output entry, and the buffers the parity leg reads
    pipe.synchronize()
    tot = torch.stack([out_nnz[0], chk[0]])
    if ctx.world > 1:
        ctx.dist.all_reduce(tot)
    alg_bytes = n * a.k * 8 + n * a.k * per * 8 + nnz_rank * 8
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    rows, (indptr, oc, ov) = res
    result = None
    if ctx.rank == 0:
        result = {
            "metric": "query cells/s end-to-end (k-NN+kernel+transfer)", "value": n * a.steps / elapsed, "unit": "query cells/s",
            "n_gpus": ctx.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "timed_region": "device-resident",
            "first_call_ms": first_call_ms,
            "config": {"workload": f"self-mapping of {n} spatial cells, d={a.d}, k={a.k}, {a.kernel} kernel, expression imputation from a "
                                   f"sparse reference.X ({G} genes, {per_all} non-zeros per cell, CSR) (BASELINE configs[4]); {ctx.world} GPU(s): "
                                   f"search {ctx.parallelism()}, product gene-column sharded ({Gs} genes / {per} non-zeros per cell on rank 0); "
                                   f"CSR product produced in {tile}-row tiles{' of a locality order of the rows' if use_order else ''}"
                                   f"{' (rows of a tile in completion order, named by a row-id vector)' if any_order else ''}, {int(tot[0].item())} "
                                   f"non-zeros in total",
                       "baseline_config": 5, "parallelism": ctx.parallelism() + f"+genes/{ctx.world}",
                       "exchange": ctx.state.describe() if (ctx.use_dist and lay.ref_shards > 1) else None,
                       "nq": n, "n_ref": n, "d": a.d, "k": a.k, "genes": G, "nnz_per_cell": per_all,
                       "genes_per_rank": Gs, "query_tile": tile, "output_nnz": int(tot[0].item()), "output_nnz_rank0": nnz_rank,
                       "data_generator": "counter-based hash streams (bench_data.py), N-independent"},
            "roofline": {"bound": "hbm", "kernel": ("spgemm_sliced (single-pass CSR x CSR product, A3: every wave owns an eighth of the output "
                                                "columns; counts, offsets, entries from one gather)" if sliced else
                                                "spgemm_rows<2> (single-pass CSR x CSR product, A3: counts, offsets, entries from one gather)")
                     if one_pass else "spgemm_rows<0> + spgemm_rows<1> (count + fill passes of the CSR x CSR product, A3)",
                     "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS, "traffic": None,
                     "kernel_ms": ms, "algorithmic_bytes": alg_bytes,
                     # the committed counter pass (profiles/r03_pmc_counters.md: 400 k cells, FETCH_SIZE x 2 + WRITE_SIZE = 0.542 x the
                     # algorithmic bytes -- the locality order turns most of the gather into L2 / Infinity-Cache hits): the fraction of the
                     # HBM peak by the counters' bytes is what says how far the kernel is from the memory system's limit
                     "frac_hbm_counters": 0.542 * achieved / PEAK_HBM_GBS if ctx.world == 1 else None,
                     "frac_hbm_counters_source": "profiles/r03_pmc_counters.md (counter bytes / algorithmic bytes = 0.542 at 400 k cells; round 2: 0.545)",
                     "kernel_ms_covers": ("all query tiles of one step incl. the locality sort and the payload index (one gather per row)" if one_pass else
                                          "both passes over all query tiles in one step (the gather is done twice: count, fill)")
                                         + ", rank 0's gene columns, mean over the timed steps"},
            "stage_ms_last_step": dict(stages, layer_spgemm=ms),
            "layer_checksum": {"value": f"{int(tot[1].item()) & 0xFFFFFFFFFFFFFFFF:016x}", "entries": int(tot[0].item()),
                               "what": "sum over ALL output entries of float32 bits x ((global gene column) mod 1000003 + 1), all ranks; "
                                       "identical at every N"},
        }
    ok = True
    xy_host = lambda: bd.fill_rows(torch, "uniform", bd.STREAM_XY, 0, n, a.d, dev, scale=2000.0).cpu().numpy()   # noqa: E731
    xy_rows = lambda ids: bd.rows_at(torch, "uniform", bd.STREAM_XY, ids, a.d, dev, scale=2000.0)                 # noqa: E731
    rec, cb_knn, ok_knn = _knn_gate(ctx, rows, xy_host, xy_rows, "k-NN+kernel")
    ok = ok and ok_knn
    if a.check > 0:
        from oracle import oracle as orc
        import scipy.sparse as sp

        orc.build()
        threads = orc.set_num_threads(__import__("bench").usable_cores())
        # rows of the LAST tile the checksum pass produced (its buffers still hold it): tile row r is mapping-matrix row
        # order[q0 + r] (or q0 + r), found in the tile's CSR at slot j with t_rows[j] == r (or j == r)
        q0, q1, b = last["q0"], last["q1"], last["b"]
        nt = q1 - q0
        ns = min(64, nt)
        tr = np.arange(0, nt, max(1, nt // ns))[:ns]
        src = (last["order"][q0:q1].cpu().numpy()[tr] if last["order"] is not None else q0 + tr).astype(np.int64)
        if any_order:
            inv_rows = np.empty(nt, dtype=np.int64)
            inv_rows[t_rows[b][:nt].cpu().numpy()] = np.arange(nt)
            slots = inv_rows[tr]
        else:
            slots = tr
        tp = t_ptr[b][:nt + 1].cpu().numpy()
        ipt = indptr.cpu().numpy()
        rows_c = np.concatenate([np.arange(ipt[i], ipt[i + 1]) for i in src])
        c_np = oc[torch.from_numpy(rows_c).to(dev)].cpu().numpy()
        v_np = ov[torch.from_numpy(rows_c).to(dev)].cpu().numpy()
        uniq, inv = np.unique(c_np, return_inverse=True)
        sel_rows = torch.from_numpy(uniq).to(dev).long()
        bc = b_cols.view(n, per)[sel_rows].cpu().numpy().ravel()
        bv = b_vals.view(n, per)[sel_rows].cpu().numpy().ravel()
        Bs = sp.csr_matrix((bv, bc, np.arange(len(uniq) + 1, dtype=np.int64) * per), shape=(len(uniq), Gs))
        lens = np.array([ipt[i + 1] - ipt[i] for i in src])
        ip2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        tq0 = time.perf_counter()
        dense = orc.spmm_csr(ip2, inv.astype(np.int32), v_np, Bs)   # [ns, Gs] float32; csr_matmat drops exact zeros
        agg_s = (time.perf_counter() - tq0) / max(len(src), 1) * (per_all / max(per, 1))   # per query row, scaled to all genes
        ok_rows = bool(tp[0] == 0 and np.all(np.diff(tp) >= 0) and tp[-1] <= t_cols[b].shape[0])
        for r, j in enumerate(slots.tolist()):
            gc = t_cols[b][tp[j]:tp[j + 1]].cpu().numpy()
            gv = t_vals[b][tp[j]:tp[j + 1]].cpu().numpy()
            wc = np.flatnonzero(dense[r])
            ok_rows = ok_rows and np.array_equal(gc, wc) and np.array_equal(gv, dense[r][wc])
        ok_all = ctx.all_true(bool(ok_rows))
        ok = ok and ok_all
        if ctx.rank == 0:
            result["parity_check"] = {"knn": rec, "rows_checked_per_rank": int(ns), "csr_rows_bit_exact": bool(ok_rows), "all_ranks_ok": ok_all,
                                      "ok": bool(ok), "checked": "rows of the last tile produced, against the CPU oracle (scipy csr_matmat restatement)"}
            cb = _combine_cpu(cb_knn, rec, agg_s, f"oracle csr_matmat restatement on {len(src)} rows x {per} of {per_all} non-zeros per cell "
                                                  f"(scaled), {threads} OpenMP threads", threads)
            if cb is not None:
                result["cpu_baseline"] = cb
    ctx.ref_lo, ctx.ref_hi = old_ref_bounds
    return result, ok
