#!/usr/bin/env python3
"""Benchmark of the CellMapper hot path on MI355X: query cells/s end-to-end (k-NN + kernel + transfer).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  With N > 1 and no ``WORLD_SIZE`` in the
environment this process only LAUNCHES: it starts ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a
child BEFORE anything touches the GPU and exits with the child's code (a process that has initialised the GPU is never
re-executed).  Under ``torch.distributed.run`` (the driver's own launch shape) every rank runs ``main_rank``: one rank
per GPU over RCCL.  Rank 0 prints ONE JSON line.

Workloads (``--config``, numbered like SURVEY.md 8d / BASELINE.json configs[1..4]):
  2 (default, the headline metric) 1.5 M query x 1.5 M reference cells, float32 standard-normal d=50, k=30, Gaussian
    kernel, one 20-class label column;
  3 the same search + obsm transfer (d_out = 2) + dense layer transfer over reference.X with 20 000 genes (A1 + A2);
  4 1.5 M query x 12 M reference, d=128, k=30, label transfer; at N > 1 the reference is sharded over ALL ranks;
  5 self-mapping of 4 M spatial cells (d=2, k=90) + expression imputation from a sparse reference.X (30 000 genes,
    3 % density), output produced in query tiles (it is ~0.9 TB as CSR and is never resident as a whole).
Inputs are resident in HBM when the timed region starts and outputs stay in HBM ("timed_region": "device-resident");
the host-arrays-in -> host-results-out time of the same workload through the drop-in host tier is reported next to it
("host_inclusive") and is never `value`.  A "step" is one full pass of the hot path over all queries.  At N > 1 the
REFERENCE rows are sharded across ranks and per-shard top-k lists are merged after one RCCL collective (strong scaling:
the problem is fixed).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, dense MFMA peaks: f32-input 157.3 TFLOP/s, f16/bf16 ~2500 TFLOP/s
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0

# per-config defaults: (nq, nref, d, k, description)
CONFIGS = {
    2: dict(nq=1_500_000, nref=1_500_000, d=50, k=30, genes=0, what="20-class label transfer (BASELINE configs[1])"),
    3: dict(nq=1_500_000, nref=1_500_000, d=50, k=30, genes=20_000,
            what="obsm (d_out=2) + dense layer X transfer (BASELINE configs[2])"),
    4: dict(nq=1_500_000, nref=12_000_000, d=128, k=30, genes=0,
            what="20-class label transfer, reference sharded over all ranks (BASELINE configs[3])"),
    5: dict(nq=4_000_000, nref=4_000_000, d=2, k=90, genes=30_000,
            what="self-mapping of spatial cells + sparse-X expression imputation (BASELINE configs[4])"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--nq", type=int, default=0, help="override the config's query count")
    ap.add_argument("--nref", type=int, default=0, help="override the config's reference count")
    ap.add_argument("--d", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--genes", type=int, default=-1, help="override the config's gene count (configs 3 and 5)")
    ap.add_argument("--density", type=float, default=0.03, help="config 5: density of the sparse reference.X")
    ap.add_argument("--tile", type=int, default=0, help="configs 3/5: query rows per output tile (0 = automatic)")
    ap.add_argument("--ncat", type=int, default=20)
    ap.add_argument("--kernel", default="gaussian")
    ap.add_argument("--metric", default="euclidean", choices=["euclidean", "cosine", "manhattan", "chebyshev"])
    ap.add_argument("--exchange", default="alltoall", choices=["alltoall", "allgather"])
    ap.add_argument("--ref-shards", type=int, default=0,
                    help="reference shards R (world = query groups x R); 0 = default (2 when world is even; the whole "
                         "world for config 4), world = pure reference sharding")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (+ --share-gpu) rehearses the N > 1 path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and take the sharded code path even with one rank (RCCL smoke test)")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--check", type=int, default=4096, help="queries verified against the CPU oracle (0 = skip)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-in -> host-out measurement")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of the self-launched ranks (0 = pick)")
    a = ap.parse_args(argv)
    c = CONFIGS[a.config]
    a.nq = a.nq or c["nq"]
    a.nref = a.nref or c["nref"]
    a.d = a.d or c["d"]
    a.k = a.k or c["k"]
    a.genes = c["genes"] if a.genes < 0 else a.genes
    return a


# ------------------------------------------------------------------------------------- launcher -------------------
def compose_launch_cmd(n_gpus: int, argv: list[str], port: int) -> list[str]:
    """The command `bench.py --gpus N` runs when it has to start its own ranks: the driver's launch shape."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def needs_launch(a, environ) -> bool:
    return a.gpus > 1 and "WORLD_SIZE" not in environ


def pick_port(a) -> int:
    if a.master_port:
        return a.master_port
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a, argv) -> int:
    """Parent of a self-launched multi-GPU run: never touches the GPU, only waits for the child."""
    import torch  # device_count() does not initialise the GPU

    n_dev = torch.cuda.device_count()
    if not a.share_gpu and n_dev < a.gpus:
        print(f"bench.py: --gpus {a.gpus} requested but only {n_dev} GPU(s) are visible", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = compose_launch_cmd(a.gpus, argv, pick_port(a))
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------- CPU legs -------------------
def cpu_baseline(ref_np, qry_np, codes_np, k, ncat, kernel, n_sample):
    """Time the CPU oracle (plain C restatement of the reference path: float64, query-blocked / cache-tiled scan, OpenMP
    over the host cores) on a bounded sample: `n_sample` queries against the FULL reference; brute force is linear in the
    number of queries.  The one-off re-layout of the reference (float64 upcast into tiles) is set-up: measured, stated,
    and excluded from the rate -- at full size it is amortised over 1.5 M queries."""
    # threads = the cores this process may actually use (cgroup CPU quota / affinity), not the host's core count: a GPU box
    # gives a one-GPU job a share of its cores, and oversubscribing them only slows the scan down
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))
    from oracle import oracle as orc

    orc.build()
    q = qry_np[:n_sample]
    t0 = time.perf_counter()
    out = orc.run_path(ref_np, q, k, kernel=kernel, labels=codes_np, n_cat=ncat)
    dt = time.perf_counter() - t0
    setup = orc.last_setup_seconds()
    return dict(value=n_sample / max(dt - setup, 1e-9), unit="query cells/s", cores=orc.num_threads(), kind="port",
                sample=f"{n_sample} queries x {ref_np.shape[0]} reference rows, d={ref_np.shape[1]}, k={k}, "
                       f"k-NN+kernel+label vote, {dt - setup:.2f} s of scan (+ {setup:.2f} s one-off reference re-layout, "
                       f"excluded)"), out


_SKLEARN_CHILD = r"""
import json, sys, time
import numpy as np
d = np.load(sys.argv[1])
ref, q, k = d["ref"], d["q"], int(d["k"])
import sklearn
from sklearn.neighbors import NearestNeighbors
nn = NearestNeighbors(n_neighbors=k, metric="euclidean").fit(ref)
t0 = time.perf_counter()
dist, idx = nn.kneighbors(q)
dt = time.perf_counter() - t0
np.save(sys.argv[2], idx)
print(json.dumps({"seconds": dt, "version": sklearn.__version__}))
"""


def sklearn_baseline(ref_np, qry_np, k, n_sample):
    """scikit-learn's own exact search (the third-party package the reference calls, knn.py:429-440) on the same sample, if
    it is importable on this box.  Runs in a CHILD process with a bounded thread count and a timeout: on many-core hosts
    sklearn's OpenMP threads calling OpenBLAS can exhaust OpenBLAS's thread table and abort the interpreter -- that must not
    take the benchmark down.  Returns (record, indices) or None."""
    import tempfile

    q = qry_np[:n_sample]
    cores = min(usable_cores(), 64)
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            fin, fout = os.path.join(td, "in.npz"), os.path.join(td, "idx.npy")
            np.savez(fin, ref=ref_np, q=q, k=np.int64(k))
            env = dict(os.environ, OMP_NUM_THREADS=str(cores), OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            r = subprocess.run([sys.executable, "-c", _SKLEARN_CHILD, fin, fout], env=env, capture_output=True, text=True, timeout=120)
            if r.returncode != 0:
                return None
            info = json.loads(r.stdout.strip().splitlines()[-1])
            idx = np.load(fout)
    except Exception:
        return None
    dt = info["seconds"]
    return {"value": n_sample / dt, "unit": "query cells/s", "cores": cores, "kind": "sklearn",
            "sample": f"sklearn {info['version']} NearestNeighbors.kneighbors (child process, {cores} OpenMP threads), {n_sample} queries x "
                      f"{ref_np.shape[0]} reference rows, d={ref_np.shape[1]}, k={k}, search only, {dt:.2f} s"}, idx


def usable_cores() -> int:
    """CPU cores available to this process: min(affinity mask, cgroup v2 / v1 CPU quota, os.cpu_count())."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:   # cgroup v2: "<quota|max> <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def host_inclusive_leg(a, ref_np, qry_np, codes_np):
    """Host arrays in -> host results out through the graph tier of the C ABI -- exactly what `CellMapper.map(obs_keys=...)`
    calls: both embeddings uploaded (cm_rep_create), search and mapping matrix kept on the device (cm_graph_knn,
    cm_graph_mapping), label codes up, labels and confidences down (cm_graph_label_vote).  PCIe copies included.
    SURVEY.md 8d's definition of the metric; reported next to `value`, never as it."""
    from cellmapper_amd import _native

    h = _native.default_handle()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        xr, yr = _native.DeviceRep(h, ref_np), _native.DeviceRep(h, qry_np)
        g = _native.DeviceGraph(h, xr, yr, a.k, a.metric)
        xr.close(); yr.close()
        g.mapping(a.kernel)
        code, conf = g.label_vote(codes_np, a.ncat)
        dt = time.perf_counter() - t0
        g.close()
        best = dt if best is None else min(best, dt)
    return {"seconds": best, "value": qry_np.shape[0] / best, "unit": "query cells/s",
            "what": "numpy in -> numpy out through the graph tier of the C ABI (cm_rep_create x2, cm_graph_knn, "
                    "cm_graph_mapping, cm_graph_label_vote), H2D/D2H included, best of 2"}


def git_rev(path):
    try:
        return subprocess.check_output(["git", "log", "-1", "--format=%h", "--", path], cwd=ROOT, text=True,
                                       stderr=subprocess.DEVNULL).strip() or None
    except Exception:
        return None


# ------------------------------------------------------------------------------------- one rank -------------------
def main_rank(a):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if a.share_gpu:
        local_rank = 0
    if a.config in (3, 5) and world > 1:
        print("bench.py: configs 3 and 5 are single-GPU workloads here (gene-column sharding needs no collective; "
              "run one process per gene slab)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from cellmapper_amd.device import DevicePipeline
    from cellmapper_amd.sharded import HipOps, grid_2d, make_groups, shard_bounds, sharded_map_labels

    pipe = DevicePipeline(local_rank)
    pipe.h.set_profiling(True)

    if a.config in (3, 5):
        import bench_wide   # configs 3 / 5: aggregation-heavy single-GPU workloads (scripts live next to bench.py)

        result = bench_wide.run(a, pipe, torch)
        print(json.dumps(result), flush=True)
        if result.get("parity_check", {}).get("ok") is False:
            sys.exit(4)
        return

    # synthetic inputs: standard-normal float32 embeddings drawn separately for reference and query
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    ref_full_rows = a.nref
    default_shards = world if a.config == 4 else None
    n_groups, ref_shards = grid_2d(world, a.ref_shards or default_shards)
    qgroup, ref_rank = rank // ref_shards, rank % ref_shards
    pg = make_groups(dist, world, n_groups, ref_shards, a.exchange) if world > 1 else None
    lo, hi = shard_bounds(ref_full_rows, ref_shards, ref_rank)
    gq_lo, gq_hi = shard_bounds(a.nq, n_groups, qgroup)  # this query group's rows
    # every rank draws the same stream and keeps its shard, so the global problem is independent of N; chunks that do
    # not touch the rank's shard only advance the generator (same element count, no [chunk, d] buffer kept)
    ref = torch.empty((hi - lo, a.d), dtype=torch.float32, device=dev)
    chunk = 250_000
    for c0 in range(0, ref_full_rows, chunk):
        c1 = min(ref_full_rows, c0 + chunk)
        blk = torch.randn((c1 - c0, a.d), generator=g, dtype=torch.float32, device=dev)
        s0, s1 = max(c0, lo), min(c1, hi)
        if s0 < s1:
            ref[s0 - lo:s1 - lo] = blk[s0 - c0:s1 - c0]
        del blk
    qry = torch.empty((gq_hi - gq_lo, a.d), dtype=torch.float32, device=dev)
    for c0 in range(0, a.nq, chunk):
        c1 = min(a.nq, c0 + chunk)
        blk = torch.randn((c1 - c0, a.d), generator=g, dtype=torch.float32, device=dev)
        s0, s1 = max(c0, gq_lo), min(c1, gq_hi)
        if s0 < s1:
            qry[s0 - gq_lo:s1 - gq_lo] = blk[s0 - c0:s1 - c0]
        del blk
    codes = torch.randint(0, a.ncat, (ref_full_rows,), generator=g, device=dev, dtype=torch.int32)
    torch.cuda.synchronize()

    ops = HipOps(pipe)
    torch.cuda.set_stream(pipe.stream)  # library kernels, torch ops and RCCL collectives share one stream

    q_bounds = [None]   # this rank's slice of its query group in the last sharded step

    def step():
        if not use_dist:
            return pipe.map_labels(ref, qry, a.k, codes, a.ncat, metric=a.metric, kernel=a.kernel)
        code, conf, qb, _, _ = sharded_map_labels(ops, dist, ref, lo, ref_full_rows, qry, a.k, codes, a.ncat,
                                                  metric=a.metric, kernel=a.kernel, exchange=a.exchange, group=pg,
                                                  ref_world=ref_shards, ref_rank=ref_rank)
        q_bounds[0] = qb
        return code, conf

    def check_overflow():
        ov = pipe.h.knn_stats().get("fallback_overflow", 0)
        if ov:
            print(f"[bench] rank {rank}: exact-fallback overflow (more than 4096 rows within a query's bound): results "
                  f"would be truncated", file=sys.stderr, flush=True)
            sys.exit(5)

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    p1_ms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    check_overflow()
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
        # HBM traffic of the dominant kernel: PMC counters cannot be read inside this run; the committed counter pass
        # (scripts/kernel_counters.sh -> profiles/phase1_hbm_traffic.json) is cited with the commit that produced it
        traffic = traffic_src = None
        tfile = os.path.join(ROOT, "profiles", "phase1_hbm_traffic.json")
        if os.path.exists(tfile) and world == 1 and a.config == 2 and a.nq == 1_500_000 and a.nref == 1_500_000:
            try:
                tj = json.load(open(tfile))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = {"file": "profiles/phase1_hbm_traffic.json", "file_commit": git_rev(tfile),
                               "kernel_source_commit": git_rev(os.path.join(ROOT, "cellmapper_amd", "csrc", "knn_phase1_f16.h")),
                               "measured_on_commit": tj.get("measured_on_commit")}
            except Exception:
                traffic = None
        result = {
            "metric": "query cells/s end-to-end (k-NN+kernel+transfer)",
            "value": value, "unit": "query cells/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16" if fb["lowd_path"] in (2, 4) else "f32",
            "data": "synthetic",
            "timed_region": "device-resident",   # inputs and outputs in HBM; host-in -> host-out is `host_inclusive`
            "config": {"workload": f"{a.nq} query x {a.nref} reference cells, d={a.d}, k={a.k}, {a.metric}, {a.kernel} kernel, "
                                   f"{CONFIGS[a.config]['what']}; {world} GPU(s) = {n_groups} query "
                                   f"group(s) x {ref_shards} reference shard(s), exchange="
                                   f"{a.exchange if ref_shards > 1 else 'none'}",
                       "baseline_config": a.config,
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
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": p1,
                         # HIP events around the filter stage on the handle's stream
                         "kernel_ms_covers": "all launches of the filter kernel in the step",
                         "flops_per_launch": flops, "executed_mfma_tflops": executed / (p1 * 1e-3) / 1e12,
                         "executed_frac_of_peak": executed / (p1 * 1e-3) / 1e12 / peak,
                         "f32_mfma_equivalent_frac": achieved / PEAK_F32_MFMA_TFLOPS},
            "stage_ms_last_step": stages,
        }
    # rehearsal of the N > 1 path on a one-GPU box (--share-gpu): every rank checks its slice of the sharded result against
    # the same queries mapped onto the FULL reference by the single-GPU pipeline (identical labels, confidences to 1e-5)
    if use_dist and (a.share_gpu or a.force_dist):
        g2 = torch.Generator(device=dev)
        g2.manual_seed(0)
        ref_full = torch.cat([torch.randn((min(ref_full_rows, c0 + chunk) - c0, a.d), generator=g2, dtype=torch.float32, device=dev)
                              for c0 in range(0, ref_full_rows, chunk)])
        qry_full = torch.cat([torch.randn((min(a.nq, c0 + chunk) - c0, a.d), generator=g2, dtype=torch.float32, device=dev)
                              for c0 in range(0, a.nq, chunk)])   # (same stream order as above)
        ql, qh = q_bounds[0]
        one = pipe.map_labels(ref_full, qry_full, a.k, codes, a.ncat, metric=a.metric, kernel=a.kernel)  # (global bandwidth)
        torch.cuda.synchronize()
        sl = slice(gq_lo + ql, gq_lo + qh)
        same = bool(torch.equal(one[0][sl], out[0])) and bool(torch.allclose(one[1][sl], out[1], rtol=1e-5, atol=1e-6))
        print(f"[rehearsal] rank {rank}/{world} ({a.backend}, exchange={a.exchange}): sharded slice [{ql}, {qh}) of group "
              f"{qgroup} vs single-GPU map of all queries: {'identical' if same else 'MISMATCH'}", file=sys.stderr, flush=True)
        if not same:
            sys.exit(3)
    # parity spot check + CPU baseline + host-inclusive leg on rank 0 at N=1 only (bounded, outside the timed region)
    parity_ok = True
    if rank == 0 and world == 1 and not a.force_dist:
        ref_np = qry_np = codes_np = None
        if a.cpu_sample > 0 or a.check > 0 or not a.no_host_leg:
            ref_np = ref.cpu().numpy()
            qry_np = qry.cpu().numpy()
            codes_np = codes.cpu().numpy()
        if (a.cpu_sample > 0 or a.check > 0) and a.metric == "euclidean":
            # bounded CPU leg: the sample shrinks with the cost of a query (n_ref x d) so that it stays ~5 s of scan
            scale = min(1.0, (1_500_000 * 50) / float(a.nref * a.d))
            ns = max(256, int(max(a.cpu_sample, a.check) * scale)) if max(a.cpu_sample, a.check) > 256 else max(a.cpu_sample, a.check)
            cb, o = cpu_baseline(ref_np, qry_np, codes_np, a.k, a.ncat, a.kernel, ns)
            if a.cpu_sample > 0:
                result["cpu_baseline"] = cb
                sk = sklearn_baseline(ref_np, qry_np, a.k, min(ns, 2048))
                if sk is not None:
                    result["cpu_baseline_sklearn"] = sk[0]
                    # (not a parity gate: sklearn breaks exact ties by heap layout; on continuous synthetic data there are none)
                    result["cpu_baseline_sklearn"]["indices_equal_oracle"] = bool(np.array_equal(sk[1], o["idx"][:sk[1].shape[0]]))
            if a.check > 0:
                # an evenly strided sample of the whole query set would need the oracle on other rows; the first `ns`
                # rows are as random as any (i.i.d. synthetic data).  The kernel bandwidth is a global statistic, so
                # labels / confidences are compared on a run over the same `ns` queries.
                code, conf, d_gpu, i_gpu = pipe.map_labels(ref, qry[:ns].contiguous(), a.k, codes, a.ncat, kernel=a.kernel,
                                                          want_graph=True)
                torch.cuda.synchronize()
                check_overflow()
                idx_ok = bool(np.array_equal(i_gpu.cpu().numpy(), o["idx"]))
                dist_ok = bool(np.array_equal(d_gpu.cpu().numpy(), o["dist"]))
                lab = float((code.cpu().numpy() == o["pred_code"]).mean())
                conf_ok = bool(np.allclose(conf.cpu().numpy(), o["conf"], rtol=1e-5, atol=1e-7))
                parity_ok = idx_ok and dist_ok and conf_ok and lab >= 0.9999
                result["parity_check"] = {"queries": ns, "indices_bit_exact": idx_ok, "distances_bit_exact": dist_ok,
                                          "label_agreement": lab, "confidences_within_1e-5": conf_ok, "ok": parity_ok}
        if a.check > 0 and a.metric != "euclidean":
            # other metrics: a small sample through the numpy restatements of the oracle (no CPU throughput leg)
            from oracle import oracle as orc
            orc.build()
            ns = min(a.check, 256 if a.metric == "cosine" else 16)
            o = orc.run_path(ref_np, qry_np[:ns], a.k, metric=a.metric, kernel=a.kernel, labels=codes_np, n_cat=a.ncat)
            code, conf, d_gpu, i_gpu = pipe.map_labels(ref, qry[:ns].contiguous(), a.k, codes, a.ncat, metric=a.metric, kernel=a.kernel,
                                                      want_graph=True)
            torch.cuda.synchronize()
            check_overflow()
            i_np, d_np = i_gpu.cpu().numpy(), d_gpu.cpu().numpy()
            idx_ok = bool(np.array_equal(i_np, o["idx"]))
            # (float32 cosine: sklearn ranks float32 values, the fused pipeline the float64 ones -- tie order at rounding level)
            dist_ok = bool(np.array_equal(d_np, o["dist"])) if a.metric != "cosine" else bool(np.allclose(d_np, o["dist"], rtol=0, atol=4e-7))
            if a.metric == "cosine" and not idx_ok:
                idx_ok = bool((np.sort(i_np, 1) == np.sort(o["idx"], 1)).mean() > 0.9999)
            conf_ok = bool(np.allclose(conf.cpu().numpy(), o["conf"], rtol=1e-5, atol=1e-7))
            lab = float((code.cpu().numpy() == o["pred_code"]).mean())
            parity_ok = idx_ok and dist_ok and conf_ok and lab >= 0.999
            result["parity_check"] = {"queries": ns, "indices_exact": idx_ok, "distances_exact": dist_ok, "label_agreement": lab,
                                      "confidences_within_1e-5": conf_ok, "ok": parity_ok}
        if not a.no_host_leg and a.config == 2:
            pipe.h.set_profiling(False)
            result["host_inclusive"] = host_inclusive_leg(a, ref_np, qry_np, codes_np)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        print("[bench] parity check against the CPU oracle FAILED", file=sys.stderr, flush=True)
        sys.exit(4)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse(argv)
    if needs_launch(a, os.environ):
        sys.exit(launch_ranks(a, argv))
    main_rank(a)


if __name__ == "__main__":
    main()
