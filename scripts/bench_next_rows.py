#!/usr/bin/env python3
"""Timings of the "next" rows of SURVEY.md 8f on the GPU: precomputed-distance ingest (host tier, PCIe included) and
expression-transfer evaluation (device tier: both matrices resident in HBM)."""
import ctypes as C, json, os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cellmapper_amd import _native
from cellmapper_amd.utils import extract_neighbors_from_distances

h = _native.default_handle()
out = {}
# ---- ingest: 4 M cells x 15 stored neighbours (spatial-graph like), rows stored unsorted
n, k = 4_000_000, 15
rng = np.random.default_rng(0)
indptr = np.arange(0, n * k + 1, k, dtype=np.int64)
indices = rng.integers(0, n, n * k).astype(np.int32)
data = rng.random(n * k)
m = sp.csr_matrix((data, indices, indptr), shape=(n, n))
for rep in range(2):
    t0 = time.perf_counter(); idx, dist = extract_neighbors_from_distances(m, include_self=True); t1 = time.perf_counter()
out["ingest"] = {"cells": n, "stored_per_row": k, "include_self": True, "seconds_host_tier": t1 - t0, "kmax": int(idx.shape[1])}
del m, idx, dist, indices, data
# ---- evaluation: 200 k cells x 20 k genes float32, resident on the device, 8 groups
n, G, ng = 200_000, 20_000, 8
dev = torch.device("cuda", 0)
a = torch.rand((n, G), device=dev); b = a * 0.5 + torch.rand((n, G), device=dev) * 0.5
grp = torch.randint(0, ng, (n,), device=dev, dtype=torch.int32)
res = torch.empty((ng + 1, G), device=dev)
torch.cuda.synchronize()
for method, mid in (("pearson", 0), ("js", 2), ("spearman", 3)):
    for groups in (0, ng):
        ts = []
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h.call("cm_expression_metrics_dev", C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, G,
                   C.c_void_p(grp.data_ptr()) if groups else None, groups, mid, C.c_void_p(res.data_ptr()))
            h.call("cm_synchronize"); ts.append(time.perf_counter() - t0)
        gb = 2 * 2 * n * G * 4 * (groups + 1) / 1e9   # two passes over both matrices per cell selection (spearman: sort-bound, not HBM)
        out[f"eval_{method}_groups{groups}"] = {"seconds": ts[-1], "GB_read": gb, "TB_per_s": gb / ts[-1] / 1e3}
print(json.dumps(out))
