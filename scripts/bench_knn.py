#!/usr/bin/env python3
"""Device-resident k-NN timing for arbitrary shapes (stage times from HIP events inside the library)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=1_000_000); ap.add_argument("--nref", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=2); ap.add_argument("--k", type=int, default=90)
    ap.add_argument("--uniform", type=float, default=2000.0); ap.add_argument("--self", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--clusters", type=int, default=0, help="draw rows from this many Gaussian clusters (centres N(0, spread^2))")
    ap.add_argument("--spread", type=float, default=5.0); ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--sorted", action="store_true", help="reference rows sorted by cluster label (worst case for streaming thresholds)")
    ap.add_argument("--ragged", action="store_true", help="clusters of unequal size (Dirichlet(0.7) weights, different on the two sides), their own spread "
                                                          "(0.3 .. 3 x sigma) and a random per-coordinate scale: closer to real embeddings than equal isotropic blobs")
    a = ap.parse_args()
    import torch
    from cellmapper_amd.device import DevicePipeline
    pipe = DevicePipeline(0); pipe.h.set_profiling(True); torch.cuda.set_stream(pipe.stream)
    g = torch.Generator(device=pipe.device); g.manual_seed(0)
    mk = (lambda n: torch.rand((n, a.d), generator=g, device=pipe.device) * a.uniform) if a.uniform > 0 else \
         (lambda n: torch.randn((n, a.d), generator=g, device=pipe.device))
    if a.clusters > 0:
        cen = torch.randn((a.clusters, a.d), generator=g, device=pipe.device) * a.spread
        if a.ragged:
            csig = torch.exp(torch.rand(a.clusters, generator=g, device=pipe.device) * 2.3 - 1.2)
            aniso = torch.exp(torch.randn(a.d, generator=g, device=pipe.device) * 0.5)
            torch.manual_seed(0)
            wts = [torch.distributions.Dirichlet(torch.full((a.clusters,), 0.7)).sample().to(pipe.device) for _ in range(2)]
        side = [0]
        def mk(n):
            if a.ragged:
                lab = torch.multinomial(wts[side[0] % 2], n, replacement=True, generator=g); side[0] += 1
                return cen[lab] + (a.sigma * csig[lab])[:, None] * torch.randn((n, a.d), generator=g, device=pipe.device) * aniso[None, :]
            lab = torch.randint(0, a.clusters, (n,), generator=g, device=pipe.device)
            if a.sorted:
                lab = torch.sort(lab).values
            return cen[lab] + torch.randn((n, a.d), generator=g, device=pipe.device) * a.sigma
    ref = mk(a.nref); qry = ref if a.self else mk(a.nq)
    for _ in range(a.reps):
        dist, idx = pipe.knn(ref, qry, a.k)
    torch.cuda.synchronize()
    st = pipe.h.stage_times(); ks = pipe.h.knn_stats()
    tot = sum(st[x] for x in ("prep", "phase1_mfma", "rerank", "fallback"))
    print(json.dumps({"nq": qry.shape[0], "n_ref": a.nref, "d": a.d, "k": a.k, "stage_ms": st, "knn_ms": tot,
                      "cells_per_s": qry.shape[0] / (tot * 1e-3), "pairs_per_s": qry.shape[0] * a.nref / (st["phase1_mfma"] * 1e-3), "stats": ks}))
main()
