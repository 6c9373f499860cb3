#!/usr/bin/env python3
"""What ONE rank does per step at N = 2 / 4 / 8, measured on one GPU with the current library (round 5, review item 2).

A rank of the layout ``G query groups x R reference shards`` (N = G * R) searches ``nq / G`` queries against ``n_ref / R``
reference rows, hands its packed lists to the exchange, merges R lists per query for its ``nq / N`` queries, and runs the
graph kernel + label vote on those.  The synthetic rows are i.i.d. (bench_data.py), so any shard is statistically every
shard: this script runs that share on GPU 0 and times, with HIP events on the library's stream, (1) the local search --
with the handle's own stage clocks: preparation, filter kernel (incl. the seed scan), re-rank, fallback -- (2) the strided
merge of R packed lists (the received buffer is made of this rank's own lists, indices shifted per part: same shape, same work), (3) bandwidth
statistics + weights + vote on the slice.  The exchange itself is not measurable on one GPU; its bytes are printed with the
time they take at a stated link rate.

Output: one JSON line per share (`--out`), and the table of profiles/r05_rank_shares.md is made from them by
scripts/summarize_rank_shares.py.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK_F16 = 2500.0   # TFLOP/s dense f16 (MI355X_MICROARCH.md)
XGMI_LINK_GBS = 153.0 * 0.5   # per-direction rate assumed for one xGMI link under an RCCL all-to-all (half the 153 GB/s headline)

SHARES = {
    # name: (config, N, G, R): nq_total / n_ref_total / d come from the config
    "c2_N1": (2, 1, 1, 1),
    "c2_N2_q1xr2": (2, 2, 1, 2), "c2_N2_q2xr1": (2, 2, 2, 1),
    "c2_N4_q2xr2": (2, 4, 2, 2), "c2_N4_q1xr4": (2, 4, 1, 4), "c2_N4_q4xr1": (2, 4, 4, 1),
    "c2_N8_q4xr2": (2, 8, 4, 2), "c2_N8_q2xr4": (2, 8, 2, 4), "c2_N8_q1xr8": (2, 8, 1, 8), "c2_N8_q8xr1": (2, 8, 8, 1),
    "c4_N8_q1xr8": (4, 8, 1, 8), "c4_N8_q2xr4": (4, 8, 2, 4), "c4_N8_q4xr2": (4, 8, 4, 2),
}
CONFIGS = {2: dict(nq=1_500_000, nref=1_500_000, d=50, k=30), 4: dict(nq=1_500_000, nref=12_000_000, d=128, k=30)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shares", default="all")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch

    import bench_data as bd
    from cellmapper_amd.device import DevicePipeline
    from cellmapper_amd.sharded import HipOps

    dev = torch.device("cuda", 0)
    pipe = DevicePipeline(0)
    pipe.h.set_profiling(True)
    torch.cuda.set_stream(pipe.stream)
    ops = HipOps(pipe)
    names = list(SHARES) if a.shares == "all" else a.shares.split(",")
    out_f = open(a.out, "a") if a.out else None
    for name in names:
        cfg, N, G, R = SHARES[name]
        c = CONFIGS[cfg]
        k, d, ncat = c["k"], c["d"], 20
        nq_g = -(-c["nq"] // G)
        nref_s = -(-c["nref"] // R)
        per = -(-nq_g // R)
        nq_pad = per * R
        ref = bd.fill_rows(torch, "normal", bd.STREAM_REF, 0, nref_s, d, dev)
        qry = bd.fill_rows(torch, "normal", bd.STREAM_QRY, 0, nq_g, d, dev)
        codes = bd.integers(torch, bd.STREAM_LABEL, 0, c["nref"], ncat, dev)
        torch.cuda.synchronize()
        ev = lambda: torch.cuda.Event(enable_timing=True)
        acc = {"search": 0.0, "repack": 0.0, "merge": 0.0, "kernel_vote": 0.0}
        stages = {}
        trace = bool(os.environ.get("RANK_SHARES_TRACE"))

        def tr(msg):   # (fault hunting: every step synchronised and announced)
            if trace:
                torch.cuda.synchronize()
                print(f"[trace] {name} it={it}: {msg}", file=sys.stderr, flush=True)

        for it in range(a.warmup + a.steps):
            e = [ev() for _ in range(5)]
            tr("search ...")
            e[0].record()
            send = ops.local_knn_packed(ref, qry, k, "euclidean", 0, nq_pad)
            e[1].record()
            tr("search done")
            st = pipe.h.stage_times()   # (synchronises: the search's own stage clocks)
            stats = pipe.h.knn_stats()
            # the received buffer of an all-to-all among R shards: [R][2][per][k]; built from this rank's own lists
            e1b = ev(); e1b.record()
            if R > 1:
                pk = per * k
                packed = send.view(2, R, pk).transpose(0, 1).contiguous().view(-1)   # (what run_alltoall does before the collective)
            else:
                packed = send.view(-1)
            e[2].record()
            if R > 1:
                # what the all-to-all delivers: R parts of [2][per][k].  Part p = this rank's own lists of query slice p with the
                # indices moved to shard p's range -- R DISTINCT lists per query, as from R real shards (the merge ranks by
                # (value, index) and requires distinct pairs: R copies of one list would leave output slots unwritten)
                recv = packed.clone()
                rv = recv.view(R, 2, per * k)
                for p_ in range(1, R):
                    rv[p_, 1] += p_ * nref_s
                torch.cuda.synchronize()
                e2b = ev(); e2b.record()
                tr("merge ...")
                dist, idx = ops.merge_packed(recv, R, 2 * per * k, 0, per * k, per, k, "euclidean")
            else:
                e2b = ev(); e2b.record()
                dist, idx = ops.merge_packed(send, 1, 2 * nq_pad * k, 0, nq_pad * k, nq_g, k, "euclidean")
            e[3].record()
            tr("merge done; moments / weights / vote ...")
            m = ops.moments(dist, idx)
            param = ops.param(m, "gaussian")
            cols, vals, cnt = ops.weights(dist, idx, c["nref"], "gaussian", param, 1e-8)
            code, conf = ops.vote(cols, vals, cnt, codes, ncat)
            e[4].record()
            torch.cuda.synchronize()
            tr("step done")
            if it >= a.warmup:
                acc["search"] += e[0].elapsed_time(e[1])
                acc["repack"] += e1b.elapsed_time(e[2])
                acc["merge"] += e2b.elapsed_time(e[3])
                acc["kernel_vote"] += e[3].elapsed_time(e[4])
                for kk, vv in st.items():
                    stages[kk] = stages.get(kk, 0.0) + vv
            del send, packed, dist, idx, cols, vals, cnt, code, conf
        n = max(a.steps, 1)
        acc = {kk: vv / n for kk, vv in acc.items()}
        stages = {kk: vv / n for kk, vv in stages.items()}
        flops = 2.0 * nq_g * nref_s * d
        p1 = stages.get("phase1_mfma", 0.0)
        send_bytes = 16.0 * nq_pad * k * (R - 1) / max(R, 1)   # what leaves the rank in the all-to-all among its R shards
        xch_ms = send_bytes / max(R - 1, 1) / (XGMI_LINK_GBS * 1e9) * 1e3 if R > 1 else 0.0   # (R - 1) links in parallel
        step_ms = acc["search"] + acc["repack"] + acc["merge"] + acc["kernel_vote"] + xch_ms
        rec = {"share": name, "config": cfg, "N": N, "layout": f"q{G}xr{R}", "queries_searched": nq_g, "reference_rows": nref_s, "d": d, "k": k,
               "queries_owned_after_merge": per, "ms": acc, "search_stages_ms": stages,
               "filter": {"kernel": {0: "other", 1: "knn_cand_f16 (512-query blocks)", 2: "knn_cand_f16q (1024-query blocks)", 3: "knn_cand_f16r",
                                     4: "knn_cand_f16w", 5: "direct"}.get(stats.get("filter_kernel", 0), "?"),
                          "reference_splits": stats["ref_splits"], "level": stats["filter_products"], "family": stats["lowd_path"],
                          "candidates_per_query": stats["candidates"], "fallback_queries": stats["fallback_queries"],
                          "ms": p1, "frac_f16_peak": flops / (p1 * 1e-3) / 1e12 / PEAK_F16 if p1 > 0 else None},
               "fixed_per_query_share_of_step": (stages.get("prep", 0) + stages.get("rerank", 0) + stages.get("fallback", 0) + acc["repack"] + acc["merge"] + acc["kernel_vote"]) / step_ms,
               "exchange": {"bytes_sent": send_bytes, "assumed_link_GBs": XGMI_LINK_GBS, "ms_at_that_rate": xch_ms},
               "step_ms_projected": step_ms, "cells_per_s_projected": c["nq"] / (step_ms * 1e-3)}
        line = json.dumps(rec)
        print(line, flush=True)
        if out_f:
            out_f.write(line + "\n"); out_f.flush()
        del ref, qry, codes
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
