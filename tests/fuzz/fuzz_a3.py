#!/usr/bin/env python3
"""Randomised cross-check of the sparse layer product A3 (run on the GPU box): random mapping matrices (ragged rows, empty
rows, degrees from 1 to 300) times random canonical CSR payloads (widths 40 .. 32 768 columns, densities 0.2 % .. 60 %, values
of both signs so that exact zeros appear) through every form of the device tier -- count + fill pair, single pass with the
barrier kernel, single pass with the wave-sliced kernel (payload index), each with and without a locality order, chained
offsets and completion order -- must give the same bytes, and those of the CPU oracle (csr_matmat's add order).  Prints one
line per case; exits non-zero on the first mismatch."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60); ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args()
    import torch
    from cellmapper_amd.device import DevicePipeline
    from oracle import oracle as orc
    orc.build()
    pipe = DevicePipeline(0); torch.cuda.set_stream(pipe.stream)
    dev = pipe.device
    rng = np.random.default_rng(a.seed)
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).astype(dt)).to(dev)
    bad = 0
    for case in range(a.cases):
        n_ref = int(rng.choice([50, 300, 2000]))
        nq = int(rng.choice([1, 70, 600, 3000]))
        k = int(rng.choice([1, 3, 15, 64, 90, 130, 300]))
        k = min(k, n_ref)
        n_out = int(rng.choice([40, 64, 500, 3000, 8191, 20000, 32768]))
        density = float(rng.choice([0.002, 0.02, 0.1, 0.6]))
        if n_out * density * k * nq > 6e7:       # keep the dense oracle rows and the outputs small
            density = 6e7 / (n_out * k * nq)
        # mapping matrix: distinct ascending columns per row, ragged
        lens = rng.integers(0, k + 1, nq)
        cols = np.concatenate([np.sort(rng.choice(n_ref, size=l, replace=False)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        vals = rng.choice(np.array([0.25, 0.5, 1.0], np.float32), size=cols.shape[0]) if case % 2 else rng.random(cols.shape[0], dtype=np.float32)
        B = sp.random(n_ref, n_out, density=density, format="csr", dtype=np.float32, random_state=rng)
        B.data = rng.choice(np.array([-2.0, -1.0, 1.0, 2.0, 0.5], np.float32), size=B.data.shape[0]) if case % 2 else B.data
        B.sort_indices()
        dense = orc.spmm_csr(indptr, cols, vals, B)
        want_rows = [(np.flatnonzero(r), r[np.flatnonzero(r)]) for r in dense]
        d = dict(indptr=t(indptr, np.int64), cols=t(cols, np.int32), vals=t(vals, np.float32), bp=t(B.indptr, np.int64),
                 bc=t(B.indices, np.int32), bv=t(B.data, np.float32))
        args = (d["indptr"], d["cols"], d["vals"], d["bp"], d["bc"], d["bv"], n_out)
        cap = int(sum(len(w[0]) for w in want_rows))
        t0 = time.perf_counter()
        ok = True
        why = ""
        order = pipe.locality_order(d["indptr"], d["cols"], n_ref)
        pidx = pipe.payload_index(d["bp"], d["bc"], n_out)
        for use_order in (None, order):
            rows = np.arange(nq) if use_order is None else use_order.cpu().numpy()
            for mode in ("pair", "barrier", "sliced", "barrier_any", "sliced_any"):
                out = (torch.full((nq + 1,), -1, dtype=torch.int64, device=dev), torch.full((max(cap, 1),), -7, dtype=torch.int32, device=dev),
                       torch.full((max(cap, 1),), -7.0, dtype=torch.float32, device=dev))
                ids = torch.full((max(nq, 1),), -1, dtype=torch.int32, device=dev) if mode.endswith("_any") else None
                pipe.spmm_csr(*args, out=out, order=use_order, single_pass=mode != "pair", out_rows=ids,
                              payload_index=pidx if mode.startswith("sliced") else None)
                pipe.synchronize()
                op, oc, ov = (x.cpu().numpy() for x in out)
                perm = ids.cpu().numpy()[:nq] if ids is not None else np.arange(nq)
                good = op[0] == 0 and op[nq] == cap and np.array_equal(np.sort(perm), np.arange(nq))
                if good:
                    for j in range(nq):
                        wc, wv = want_rows[rows[perm[j]]]
                        if not (np.array_equal(oc[op[j]:op[j + 1]], wc) and np.array_equal(ov[op[j]:op[j + 1]], wv)):
                            good = False
                            break
                if not good:
                    ok = False
                    why += f" {mode}{'+order' if use_order is not None else ''}"
        print(json.dumps(dict(case=case, n_ref=n_ref, nq=nq, k=k, n_out=n_out, density=round(density, 5), out_nnz=cap, ok=ok,
                              failed=why.strip(), ms=round((time.perf_counter() - t0) * 1e3, 1))), flush=True)
        if not ok:
            bad += 1
            break
    print("FAILED" if bad else f"all {a.cases} cases ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
