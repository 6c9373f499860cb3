#!/usr/bin/env python3
"""Roofline measurement of the aggregation kernels (A0/A1/A2) on device-resident data.

A2: out[nq, G] = M[nq, n_ref] (k non-zeros per row, float32) @ X[n_ref, G] (float32, dense), BASELINE config 3 shape
(G = 20 000 genes, k = 30) at a query/reference count that fits next to the payload.  Algorithmic bytes per query:
k*8 (weights + columns) + k*G*4 (gathered rows) + G*4 (output row)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=200_000)
    ap.add_argument("--nref", type=int, default=400_000)
    ap.add_argument("--genes", type=int, default=20_000)
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    from cellmapper_amd.device import DevicePipeline

    pipe = DevicePipeline(0)
    dev = pipe.device
    torch.cuda.set_stream(pipe.stream)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    X = torch.rand((a.nref, a.genes), generator=g, device=dev, dtype=torch.float32)
    cols = torch.randint(0, a.nref, (a.nq, a.k), generator=g, device=dev, dtype=torch.int32)
    cols, _ = torch.sort(cols, dim=1)
    vals = torch.rand((a.nq, a.k), generator=g, device=dev, dtype=torch.float32)
    vals = vals / vals.sum(1, keepdim=True)
    indptr = (torch.arange(a.nq + 1, device=dev, dtype=torch.int64) * a.k)
    codes = torch.randint(0, 20, (a.nref,), generator=g, device=dev, dtype=torch.int32)
    umap = torch.randn((a.nref, 2), generator=g, device=dev, dtype=torch.float32)
    res = {}
    for name, fn, bytes_per_q in (
        ("A2_dense_layer", lambda: pipe.spmm_dense(indptr, cols.view(-1), vals.view(-1), X), a.k * 8 + a.k * a.genes * 4 + a.genes * 4),
        ("A1_obsm_d2", lambda: pipe.spmm_dense(indptr, cols.view(-1), vals.view(-1), umap), a.k * 8 + a.k * 2 * 4 + 2 * 4),
        ("A0_label_vote", lambda: pipe.label_vote(indptr, cols.view(-1), vals.view(-1), codes, 20), a.k * 12 + 8),
    ):
        out = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(pipe.stream)
        for _ in range(a.reps):
            out = fn()
        e1.record(pipe.stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        gbs = bytes_per_q * a.nq / (ms * 1e-3) / 1e9
        res[name] = {"ms": ms, "algorithmic_GBps": gbs, "frac_of_8TBps": gbs / 8000.0, "bytes_per_query": bytes_per_q}
        del out
    print(json.dumps({"nq": a.nq, "n_ref": a.nref, "genes": a.genes, "k": a.k, "kernels": res}))


if __name__ == "__main__":
    main()
