#!/usr/bin/env python3
"""Time of the exact scan (`knn_exact_scan`: shapes outside every filter family) next to the widest filtered shape."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from cellmapper_amd import _native
    h = _native.Handle(0)
    rng = np.random.default_rng(0)
    cases = [(2047, 100_000, 20_000, 30, "euclidean"), (2048, 100_000, 20_000, 30, "euclidean"), (4096, 50_000, 20_000, 30, "euclidean"),
             (2048, 100_000, 20_000, 30, "cosine"), (128, 100_000, 20_000, 30, "manhattan"), (132, 100_000, 20_000, 30, "manhattan"),
             (200, 100_000, 100_000, 30, "manhattan"), (50, 200_000, 20_000, 2000, "euclidean")]
    if len(sys.argv) > 1:
        cases = [c for i, c in enumerate(cases) if str(i) in sys.argv[1:]]
    for d, n_ref, nq, k, metric in cases:
        ref = rng.standard_normal((n_ref, d), dtype=np.float32)
        qry = rng.standard_normal((nq, d), dtype=np.float32)
        xr, yr = _native.DeviceRep(h, ref), _native.DeviceRep(h, qry)
        ts = []
        for rep in range(2):
            t0 = time.perf_counter()
            g = _native.DeviceGraph(h, xr, yr, k, metric=metric)
            h.synchronize()
            ts.append(time.perf_counter() - t0)
        st = h.knn_stats()
        print(json.dumps(dict(d=d, n_ref=n_ref, nq=nq, k=k, metric=metric, seconds=min(ts), path=st["lowd_path"],
                              pair_values_per_s=n_ref * nq / min(ts), flop_rate_T=2.0 * n_ref * nq * d / min(ts) / 1e12)), flush=True)
        del g, xr, yr


main()
