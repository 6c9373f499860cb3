#!/usr/bin/env python3
"""Randomised cross-check of the metric pass-through (run on the GPU box): manhattan, chebyshev, canberra and braycurtis neighbours and distances
of the GPU search against the CPU oracle (sklearn's DistanceMetric arithmetic, pinned on goldens) over random widths 1..400 (beyond 128: the exact scan),
k, dtypes, scales, offsets and duplicated rows -- bit for bit.  Prints one line per case; exits non-zero on a mismatch."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()
    from cellmapper_amd import _native
    from oracle import oracle as orc
    h = _native.default_handle()
    rng = np.random.default_rng(a.seed)
    bad = 0
    for case in range(a.cases):
        metric = str(rng.choice(["manhattan", "chebyshev", "canberra", "braycurtis", "canberra", "braycurtis", "manhattan", "chebyshev",   # (round 4: sklearn's two ratio metrics)
                                 "hamming", "jaccard", "dice", "rogerstanimoto", "russellrao", "sokalmichener", "sokalsneath", "yule"]))   # (and the counting metrics)
        counting = metric in orc.COUNT_METRICS
        d = int(rng.choice([1, 2, 3, 5, 10, 15, 16, 17, 24, 33, 50, 61, 64, 65, 100, 128, 129, 160, 257, 400]))
        n_ref = int(rng.choice([40, 700, 4097, 9000]))
        nq = int(rng.choice([1, 33, 500]))
        k = int(min(n_ref, rng.choice([1, 5, 15, 30, 64, 100, 100, 450, 1200])))   # (k > ~190: the exact scan)
        dt = np.float64 if rng.random() < 0.4 else np.float32
        scale = float(rng.choice([1e-3, 1.0, 1e3]))
        offset = float(rng.choice([0.0, 0.0, 1e4]))
        ref = (rng.standard_normal((n_ref, d)) * scale + offset).astype(dt)
        qry = (rng.standard_normal((nq, d)) * scale + offset).astype(dt)
        if n_ref >= 100 and rng.random() < 0.5:      # duplicated reference rows: exact ties
            ref[10:10 + int(rng.integers(2, 60))] = ref[10]
            qry[0] = ref[10]
        if metric in ("canberra", "braycurtis") and rng.random() < 0.5:   # exact zeros: the 0 / 0 rule, zero rows
            ref[rng.random(ref.shape) < 0.2] = 0; qry[rng.random(qry.shape) < 0.2] = 0
            if n_ref > 5: ref[3] = 0
            qry[-1] = 0
        if rng.random() < 0.3:                         # integer-valued coordinates: masses of tied distances
            ref = np.round(ref / scale * 2).astype(dt); qry = np.round(qry / scale * 2).astype(dt)
        if counting:                                   # small integers (hamming) / sparse non-zero patterns with all-false rows
            if metric == "hamming":
                ref = rng.integers(0, 3, (n_ref, d)).astype(dt); qry = rng.integers(0, 3, (nq, d)).astype(dt)
            else:
                dens = float(rng.choice([0.05, 0.3, 0.7]))
                ref = ((rng.random((n_ref, d)) < dens) * rng.uniform(0.5, 3.0, (n_ref, d))).astype(dt)
                qry = ((rng.random((nq, d)) < dens) * rng.uniform(0.5, 3.0, (nq, d))).astype(dt)
                ref[1] = 0; qry[-1] = 0
        t0 = time.perf_counter()
        try:
            dist, idx = h.knn(ref, qry, k, metric=metric)
            od, oi = orc.knn_count_family(ref, qry, k, metric) if counting else orc.knn_minkowski_family(ref, qry, k, metric)
            ok = bool(np.array_equal(idx, oi) and np.array_equal(dist, od, equal_nan=True))
            note = ""
        except NotImplementedError as e:               # more than 4 096 rows tied at a k-th distance: refused, never wrong
            ok, note = "overflow" in str(e), "overflow"
        print(json.dumps(dict(case=case, metric=metric, d=d, n_ref=n_ref, nq=nq, k=k, f64=dt == np.float64, scale=scale, offset=offset,
                              ok=ok, note=note, fb=h.knn_stats()["fallback_queries"], ms=round((time.perf_counter() - t0) * 1e3, 1))), flush=True)
        if not ok:
            bad += 1
    print("FAILED" if bad else f"all {a.cases} cases ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
