#!/usr/bin/env python3
"""Randomised cross-check of the staged search's optional machinery (run on the GPU box): for random shapes around the
size thresholds, metrics, dtypes and cluster structures, the default search (bucket pruning, seeded thresholds, hints)
must return bit-identical neighbours and distances to the search with all of it switched off
(CELLMAP_NO_IVF / CELLMAP_NO_SEED / CELLMAP_NO_HINT / CELLMAP_Q4=0; even cases force the four-accumulator filter).  Prints one line per case; exits non-zero on the first mismatch."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40); ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--wide", action="store_true", help="round 5: bias the draw towards the K = 128 family's pruned stages (62 <= d <= 128, clustered, >= 66 000 queries)")
    ap.add_argument("--clustered", action="store_true", help="round 5: bias the draw towards pruned main stages of the K <= 64 family (30 - 60 clusters, >= 131 072 queries)")
    a = ap.parse_args()
    import torch
    from cellmapper_amd import _native
    from cellmapper_amd.device import DevicePipeline
    pipe = DevicePipeline(0); torch.cuda.set_stream(pipe.stream)
    rng = np.random.default_rng(a.seed)
    g = torch.Generator(device=pipe.device); g.manual_seed(a.seed)
    bad = 0
    for case in range(a.cases):
        d = int(rng.choice([16, 20, 24, 32, 40, 50, 61, 64, 100, 128]))
        k = int(rng.choice([1, 5, 10, 15, 30, 50, 64, 100]))
        n_ref = int(rng.choice([70_000, 131_072, 140_000, 200_000, 300_000, 500_000]))
        nq = int(rng.choice([3_000, 30_000, 66_000, 100_000, 131_072, 135_000, 200_000, 300_000]))
        metric = "cosine" if rng.random() < 0.25 else "euclidean"
        dt = torch.float64 if rng.random() < 0.25 else torch.float32
        ncl = int(rng.choice([0, 0, 8, 40, 200]))
        sigma = float(rng.choice([0.05, 0.3, 1.0, 2.0]))
        if a.wide:
            d = int(rng.choice([62, 64, 72, 90, 100, 111, 128]))
            n_ref = int(rng.choice([131_072, 200_000, 300_000, 500_000]))
            nq = int(rng.choice([66_000, 100_000, 131_072, 200_000, 300_000]))
            ncl = int(rng.choice([8, 40, 40, 200]))
            sigma = float(rng.choice([0.05, 0.3, 0.6, 1.0]))
            k = int(rng.choice([1, 10, 30, 50, 100]))
        if a.clustered:
            d = int(rng.choice([16, 20, 24, 32, 40, 50, 61]))
            n_ref = int(rng.choice([200_000, 300_000, 500_000]))
            nq = int(rng.choice([131_072, 200_000, 300_000]))
            ncl = int(rng.choice([30, 40, 60]))
            sigma = float(rng.choice([0.1, 0.3, 0.6, 1.0]))
            k = int(rng.choice([1, 5, 10, 30, 50, 100]))
        # --clustered, odd cases: clusters of unequal size (Dirichlet weights), their own spread (0.3 .. 3 x sigma) and a random
        # per-coordinate scale (anisotropic), queries drawn with OTHER weights than the reference
        ragged = a.clustered and case % 2 == 1
        def mk(n, w=None):
            x = torch.randn((n, d), generator=g, device=pipe.device, dtype=torch.float32)
            if ncl:
                if ragged:
                    lab = torch.multinomial(w, n, replacement=True, generator=g)
                    x = cen[lab] + (sigma * csig[lab])[:, None] * x * aniso[None, :]
                else:
                    lab = torch.randint(0, ncl, (n,), generator=g, device=pipe.device)
                    x = cen[lab] + sigma * x
            return x.to(dt)
        cen = torch.randn((max(ncl, 1), d), generator=g, device=pipe.device) * 5.0
        if ragged:
            csig = torch.exp(torch.rand(ncl, generator=g, device=pipe.device) * 2.3 - 1.2)
            aniso = torch.exp(torch.randn(d, generator=g, device=pipe.device) * 0.5)
            wr = torch.distributions.Dirichlet(torch.full((ncl,), 0.7)).sample().to(pipe.device)
            wq = torch.distributions.Dirichlet(torch.full((ncl,), 0.7)).sample().to(pipe.device)
        ref, qry = (mk(n_ref, wr), mk(nq, wq)) if ragged else (mk(n_ref), mk(nq))
        for v in ("CELLMAP_NO_IVF", "CELLMAP_NO_SEED", "CELLMAP_NO_HINT"):
            os.environ.pop(v, None)
        os.environ["CELLMAP_Q4"] = "2" if case % 2 == 0 else "1"   # the four-accumulator filter wherever it exists / by its size rule
        # round 5: pass B of a pruned stage chained into one list per block (the rule needs >= 1 024 blocks: forced in two cases of three)
        # ... and, in the main stage, no pass A at all (single pass from sample-scan seeds; +512 keeps pass A)
        if case % 3 != 2:
            os.environ["CELLMAP_DBG"] = "256" if case % 3 == 0 else "768"
        else:
            os.environ.pop("CELLMAP_DBG", None)
        _native.reload_knobs()   # (the library latches the knobs in cm_create)
        t0 = time.perf_counter()
        d1, i1 = pipe.knn(ref, qry, k, metric=metric); pipe.h.call("cm_synchronize")
        t1 = time.perf_counter()
        st1 = pipe.h.knn_stats()
        for v in ("CELLMAP_NO_IVF", "CELLMAP_NO_SEED", "CELLMAP_NO_HINT"):
            os.environ[v] = "1"
        os.environ["CELLMAP_Q4"] = "0"
        os.environ.pop("CELLMAP_DBG", None)
        _native.reload_knobs()
        d2, i2 = pipe.knn(ref, qry, k, metric=metric); pipe.h.call("cm_synchronize")
        t2 = time.perf_counter()
        st2 = pipe.h.knn_stats()
        ok = bool(torch.equal(i1, i2)) and bool(torch.equal(d1, d2))
        if not ok and metric == "cosine":   # float32 cosine distances: ties at rounding level may swap equal-valued neighbours
            ok = bool(torch.equal(d1, d2)) and bool((torch.sort(i1, 1).values == torch.sort(i2, 1).values).float().mean() > 0.9999)
        print(json.dumps(dict(case=case, d=d, k=k, n_ref=n_ref, nq=nq, metric=metric, f64=dt == torch.float64, clusters=ncl,
                              sigma=sigma, ragged=bool(a.clustered and case % 2 == 1), ok=ok, ms_default=round((t1 - t0) * 1e3, 1), ms_plain=round((t2 - t1) * 1e3, 1),
                              level=st1["filter_products"], pruned_pct=st1["tiles_visited_pct"], passes=st1["pruning_passes"], kernel=st1["filter_kernel"], fb=st1["fallback_queries"],
                              fb_plain=st2["fallback_queries"])), flush=True)
        if not ok:
            bad += 1
            nbad = int((i1 != i2).any(dim=1).sum())
            print("MISMATCH rows:", nbad, flush=True)
        del ref, qry, d1, i1, d2, i2
    sys.exit(1 if bad else 0)


main()
