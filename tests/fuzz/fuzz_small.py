#!/usr/bin/env python3
"""Randomised cross-check on SMALL shapes (run on the GPU box): reference sets of 1 .. 6 000 rows, 1 .. 1 500 queries, widths
1 .. 5 000, k from 1 to the reference size (capped at 200; now and then up to 3 000: the exact scan), float32 / float64, duplicated rows, all five metrics -- the host tier's
cm_knn against the CPU oracle, bit for bit (float32 cosine: sklearn's float32 arithmetic, compared at 4e-7 with tie sets).
The large-shape fuzzers never visit these corners (k close to n_ref, one query, one reference row per tile ...)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=120); ap.add_argument("--seed", type=int, default=9)
    a = ap.parse_args()
    from cellmapper_amd import _native
    from oracle import oracle as orc
    h = _native.default_handle()
    rng = np.random.default_rng(a.seed)
    bad = 0
    for case in range(a.cases):
        metric = str(rng.choice(["euclidean", "euclidean", "cosine", "manhattan", "chebyshev", "sqeuclidean", "correlation", "canberra", "braycurtis"]))
        d = int(rng.choice([1, 2, 3, 7, 15, 16, 31, 50, 61, 62, 64, 65, 96, 128, 129, 200, 259, 260, 300, 512, 777, 1024, 2047, 2048, 2600, 5000]))
        if metric in ("manhattan", "chebyshev", "canberra", "braycurtis") and d > 600:
            d = int(rng.choice([129, 131, 200, 600]))      # (beyond 128 columns: the exact scan)
        n_ref = int(rng.choice([1, 2, 31, 129, 513, 2000, 6000]))
        nq = int(rng.choice([1, 2, 63, 700, 1500]))
        if d > 2047:
            nq = min(nq, 300)
        kmax = min(n_ref, 3000 if rng.random() < 0.08 else 200)      # (large k: beyond the filters' candidate buffers)
        k = int(rng.integers(1, kmax + 1)) if rng.random() < 0.5 else int(min(kmax, rng.choice([1, 15, 30])))
        dt = np.float64 if rng.random() < 0.35 else np.float32
        ref = rng.standard_normal((n_ref, d)).astype(dt)
        qry = rng.standard_normal((nq, d)).astype(dt)
        if n_ref >= 31 and rng.random() < 0.4:
            m = int(rng.integers(2, min(n_ref, 40)))
            ref[3:3 + m] = ref[3]
            qry[0] = ref[3]
        t0 = time.perf_counter()
        note = ""
        try:
            dist, idx = h.knn(ref, qry, k, metric=metric)
            if metric == "euclidean":
                od, oi = orc.knn_euclidean(ref, qry, k)
            elif metric == "sqeuclidean":
                od, oi = orc.knn_euclidean(ref, qry, k, squared=True)
            elif metric == "cosine":
                od, oi = orc.knn_cosine(ref, qry, k)
            elif metric == "correlation":
                od, oi = orc.knn_correlation(ref, qry, k)
            else:
                od, oi = orc.knn_minkowski_family(ref, qry, k, metric)
            if metric == "correlation" and d <= 2:
                ok, note = True, "d <= 2: degenerate (constant rows: scipy nan; d = 2: every distance is 0 or 2 up to rounding)"
            elif metric in ("cosine", "correlation"):
                tol = 1e-13 if metric == "correlation" else (4e-7 if dt == np.float32 else 1e-12)
                ok = bool(np.allclose(dist, od, rtol=0, atol=tol))
                if ok and not np.array_equal(idx, oi):   # rounding-level ties: same neighbour sets up to the k-th value's ties
                    dd = np.abs(np.take_along_axis(od, np.argsort(oi, 1), 1) - np.take_along_axis(dist.astype(od.dtype), np.argsort(idx, 1), 1))
                    ok = bool((np.sort(idx, 1) == np.sort(oi, 1)).mean() > 0.98)
                    note = "cosine tie order"
            else:
                ok = bool(np.array_equal(idx, oi) and np.array_equal(dist, od))
        except NotImplementedError as e:
            ok, note = ("too large" in str(e) or "overflow" in str(e)), str(e)[:60]
        print(json.dumps(dict(case=case, metric=metric, d=d, n_ref=n_ref, nq=nq, k=k, f64=dt == np.float64, ok=ok, note=note,
                              ms=round((time.perf_counter() - t0) * 1e3, 1))), flush=True)
        if not ok:
            bad += 1
    print("FAILED" if bad else f"all {a.cases} cases ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
