#!/usr/bin/env python3
"""Benchmark of the CellMapper hot path on MI355X: query cells/s end-to-end (k-NN + kernel + label transfer).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` (N>1 is launched through
``torch.distributed.run``, one rank per GPU over RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): 1.5 M query x 1.5 M reference cells, synthetic float32 standard-normal
embeddings d=50, k=30, Gaussian kernel, one 20-class label column.  Inputs are resident in HBM when the timed
region starts; outputs (labels, confidences) stay in HBM.  A "step" is one full pass of the hot path over all
queries.  At N>1 the REFERENCE rows are sharded across ranks and per-shard top-k lists are merged after one RCCL
exchange (strong scaling: the total work is fixed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, dense MFMA peaks: f32-input 157.3 TFLOP/s, f16/bf16 ~2500 TFLOP/s
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nq", type=int, default=1_500_000)
    ap.add_argument("--nref", type=int, default=1_500_000)
    ap.add_argument("--d", type=int, default=50)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--ncat", type=int, default=20)
    ap.add_argument("--kernel", default="gaussian")
    ap.add_argument("--exchange", default="alltoall", choices=["alltoall", "allgather"])
    ap.add_argument("--ref-shards", type=int, default=0,
                    help="reference shards R (world = query groups x R); 0 = default (2 when world is even), world = pure reference sharding")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (+ --share-gpu) rehearses the N > 1 path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--cpu-sample", type=int, default=768, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--check", type=int, default=256, help="queries verified against the CPU oracle (0 = skip)")
    return ap.parse_args()


def cpu_baseline(ref_np, qry_np, codes_np, k, ncat, kernel, n_sample):
    """Time the CPU oracle (plain C restatement of the reference path, OpenMP over host cores) on a bounded sample:
    `n_sample` queries against the FULL reference; brute force is linear in the number of queries."""
    from oracle import oracle as orc

    orc.build()
    q = qry_np[:n_sample]
    t0 = time.perf_counter()
    out = orc.run_path(ref_np, q, k, kernel=kernel, labels=codes_np, n_cat=ncat)
    dt = time.perf_counter() - t0
    return dict(value=n_sample / dt, unit="query cells/s", cores=orc.num_threads(), kind="port",
                sample=f"{n_sample} queries x {ref_np.shape[0]} reference rows, d={ref_np.shape[1]}, k={k}, "
                       f"k-NN+kernel+label vote, {dt:.2f} s"), out


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world
    if a.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from cellmapper_amd.device import DevicePipeline
    from cellmapper_amd.sharded import HipOps, grid_2d, make_groups, shard_bounds, sharded_map_labels

    pipe = DevicePipeline(local_rank)
    pipe.h.set_profiling(True)

    # synthetic inputs: standard-normal float32 embeddings drawn separately for reference and query
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    ref_full_rows = a.nref
    n_groups, ref_shards = grid_2d(world, a.ref_shards or None)
    qgroup, ref_rank = rank // ref_shards, rank % ref_shards
    pg = make_groups(dist, world, n_groups, ref_shards) if world > 1 else None
    lo, hi = shard_bounds(ref_full_rows, ref_shards, ref_rank)
    gq_lo, gq_hi = shard_bounds(a.nq, n_groups, qgroup)  # this query group's rows
    # every rank draws the same stream and keeps its shard, so the global problem is independent of N
    ref = torch.empty((hi - lo, a.d), dtype=torch.float32, device=dev)
    chunk = 250_000
    for c0 in range(0, ref_full_rows, chunk):
        c1 = min(ref_full_rows, c0 + chunk)
        blk = torch.randn((c1 - c0, a.d), generator=g, dtype=torch.float32, device=dev)
        s0, s1 = max(c0, lo), min(c1, hi)
        if s0 < s1:
            ref[s0 - lo:s1 - lo] = blk[s0 - c0:s1 - c0]
    qry_all = torch.randn((a.nq, a.d), generator=g, dtype=torch.float32, device=dev)
    qry = qry_all[gq_lo:gq_hi].contiguous() if n_groups > 1 else qry_all
    del qry_all
    codes = torch.randint(0, a.ncat, (ref_full_rows,), generator=g, device=dev, dtype=torch.int32)
    torch.cuda.synchronize()

    ops = HipOps(pipe)
    torch.cuda.set_stream(pipe.stream)  # library kernels, torch ops and RCCL collectives share one stream

    q_bounds = [None]   # this rank's slice of its query group in the last sharded step

    def step():
        if world == 1:
            return pipe.map_labels(ref, qry, a.k, codes, a.ncat, kernel=a.kernel)
        code, conf, qb, _, _ = sharded_map_labels(ops, dist, ref, lo, ref_full_rows, qry, a.k, codes, a.ncat,
                                                  kernel=a.kernel, exchange=a.exchange, group=pg, ref_world=ref_shards,
                                                  ref_rank=ref_rank)
        q_bounds[0] = qb
        return code, conf

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    p1_ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # per-kernel time of the dominant kernel, from HIP events recorded on the launch stream (last step)
    stages = pipe.h.stage_times()
    p1_ms.append(stages["phase1_mfma"])
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    fb = pipe.h.knn_stats()

    result = None
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = a.nq * a.steps / elapsed
        flops = 2.0 * qry.shape[0] * (hi - lo) * a.d  # algorithmic flops of the distance contraction on this rank
        if fb.get("tiles_visited_pct", 0):   # pruned scan: only the visited pairs are work the kernel did
            flops *= fb["tiles_visited_pct"] / 100.0
        p1 = float(np.mean(p1_ms))
        achieved = flops / (p1 * 1e-3) / 1e12
        # knn_stats[3]: 1 = f32-input MFMA kernel, 2 = split-fp16 MFMA (K<=64), 3 = low-d VALU, 4 = split-fp16 MFMA (K=128)
        f16_path = fb["lowd_path"] in (2, 4)
        peak = PEAK_F16_MFMA_TFLOPS if f16_path else PEAK_F32_MFMA_TFLOPS
        kname = ("knn_cand_f16 (split-fp16 v_mfma_f32_32x32x16_f16 distance GEMM + fused top-k filter)" if f16_path
                 else "knn_cand_mfma (v_mfma_f32_32x32x2_f32 distance GEMM + fused top-k filter)")
        # the split-fp16 kernel executes 1-3 MFMA products over the feature dimension padded to a multiple of 16 (+3 slots)
        kpad = 128 if fb["lowd_path"] == 4 else (32 if a.d + 3 <= 32 else 64)
        lvl = fb.get("filter_products", 3)   # level the query set started at: 1-3 split products, 4 = direct VALU filter,
        nprod = 1 if lvl == 5 else min(3, max(1, lvl))   # 5 = bucket-centred hi*hi
        executed = (nprod * 2.0 * qry.shape[0] * (hi - lo) * kpad if f16_path
                    else 2.0 * qry.shape[0] * (hi - lo) * (4 * ((a.d + 4) // 4)))
        if fb.get("tiles_visited_pct", 0):
            executed *= fb["tiles_visited_pct"] / 100.0
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "phase1_hbm_traffic.json")
        if os.path.exists(tfile) and world == 1 and a.nq == 1_500_000 and a.nref == 1_500_000:
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "query cells/s end-to-end (k-NN+kernel+transfer)",
            "value": value, "unit": "query cells/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16" if fb["lowd_path"] in (2, 4) else "f32",
            "data": "synthetic",
            "config": {"workload": f"{a.nq} query x {a.nref} reference cells, d={a.d}, k={a.k}, {a.kernel} kernel, "
                                   f"{a.ncat}-class label transfer (BASELINE configs[1]); {world} GPU(s) = {n_groups} query "
                                   f"group(s) x {ref_shards} reference shard(s), exchange="
                                   f"{a.exchange if ref_shards > 1 else 'none'}",
                       "parallelism": f"q{n_groups}xr{ref_shards}",
                       "nq": a.nq, "n_ref": a.nref, "d": a.d, "k": a.k, "kernel": a.kernel,
                       "fallback_queries_last_step": fb["fallback_queries"], "candidates_per_query": fb["candidates"],
                       "filter_products": fb.get("filter_products", 0),
                       # bucket pruning of the staged search: per cent of the (query block, reference tile) pairs the
                       # filter visited; 0 = full scan (what the roofline's brute-force flop count assumes -- i.i.d.
                       # synthetic rows have no cluster structure to prune)
                       "tiles_visited_pct": fb.get("tiles_visited_pct", 0)},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "kernel_ms": p1,
                         # HIP events around the filter stage on the handle's stream: for large sets the stage is two
                         # launches of this kernel (sample scan that seeds the thresholds, 1/64 of the rows, + main scan)
                         "kernel_ms_covers": "all launches of the filter kernel in the step (threshold-seeding sample scan + main scan)",
                         "flops_per_launch": flops, "executed_mfma_tflops": executed / (p1 * 1e-3) / 1e12,
                         "executed_frac_of_peak": executed / (p1 * 1e-3) / 1e12 / peak,
                         "f32_mfma_equivalent_frac": achieved / PEAK_F32_MFMA_TFLOPS},
            "stage_ms_last_step": stages,
        }
    # rehearsal of the N > 1 path on a one-GPU box (--share-gpu): every rank checks its slice of the sharded result against
    # the same queries mapped onto the FULL reference by the single-GPU pipeline (identical labels, confidences to 1e-5)
    if world > 1 and a.share_gpu:
        g2 = torch.Generator(device=dev)
        g2.manual_seed(0)
        ref_full = torch.randn((ref_full_rows, a.d), generator=g2, dtype=torch.float32, device=dev) if ref_full_rows <= 250_000 else \
            torch.cat([torch.randn((min(ref_full_rows, c0 + 250_000) - c0, a.d), generator=g2, dtype=torch.float32, device=dev)
                       for c0 in range(0, ref_full_rows, 250_000)])
        qry_full = torch.randn((a.nq, a.d), generator=g2, dtype=torch.float32, device=dev)   # (same stream order as above)
        ql, qh = q_bounds[0]
        one = pipe.map_labels(ref_full, qry_full, a.k, codes, a.ncat, kernel=a.kernel)   # (the kernel bandwidth is global)
        torch.cuda.synchronize()
        sl = slice(gq_lo + ql, gq_lo + qh)
        same = bool(torch.equal(one[0][sl], out[0])) and bool(torch.allclose(one[1][sl], out[1], rtol=1e-5, atol=1e-6))
        print(f"[rehearsal] rank {rank}: sharded slice [{ql}, {qh}) of group {qgroup} vs single-GPU map of all queries: "
              f"{'identical' if same else 'MISMATCH'}", file=sys.stderr, flush=True)
        if not same:
            sys.exit(3)
    # parity spot check + CPU baseline on rank 0 at N=1 only (bounded sample, outside the timed region)
    if rank == 0 and world == 1 and (a.cpu_sample > 0 or a.check > 0):
        ref_np = ref.cpu().numpy()
        qry_np = qry.cpu().numpy()
        codes_np = codes.cpu().numpy()
        ns = max(a.cpu_sample, a.check)
        cb, o = cpu_baseline(ref_np, qry_np, codes_np, a.k, a.ncat, a.kernel, ns)
        if a.cpu_sample > 0:
            result["cpu_baseline"] = cb
        if a.check > 0:
            code, conf, d_gpu, i_gpu = pipe.map_labels(ref, qry[:ns].contiguous(), a.k, codes, a.ncat, kernel=a.kernel,
                                                      want_graph=True)
            torch.cuda.synchronize()
            idx_ok = bool(np.array_equal(i_gpu.cpu().numpy(), o["idx"]))
            dist_ok = bool(np.array_equal(d_gpu.cpu().numpy(), o["dist"]))
            lab = float((code.cpu().numpy() == o["pred_code"]).mean())
            result["parity_check"] = {"queries": ns, "indices_bit_exact": idx_ok, "distances_bit_exact": dist_ok,
                                      "label_agreement": lab}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
