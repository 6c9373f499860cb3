#!/usr/bin/env python3
"""Randomised cross-check of the graph kernel and the dense transfers (run on the GPU box): random neighbour lists (k from 1
to 200, padded -1 / inf entries, duplicated columns, unsorted columns, one row, thousands of rows) through
cm_mapping_matrix (K1-K3, every kernel), cm_label_vote (A0) and cm_spmm_dense (A0n / A1 / A2: float32 and float64 payloads,
1 .. 300 columns) against the CPU oracle: identical CSR structure, float32 weights to 2e-7, float64 connectivities to 1e-14,
transfers to 1e-6.  Prints one line per case; exits non-zero on a mismatch."""
import os
os.environ.setdefault("OMP_NUM_THREADS", "16")   # (a GPU box gives one GPU 16 CPUs: the oracle's OpenMP default of one thread per host core -- 256 -- gets the
                                                 #  process throttled, and the HIP runtime's copies then stall for a minute at a time)
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100); ap.add_argument("--seed", type=int, default=13)
    a = ap.parse_args()
    from cellmapper_amd import _native
    from oracle import oracle as orc
    h = _native.default_handle()
    rng = np.random.default_rng(a.seed)
    bad = 0
    for case in range(a.cases):
        k = int(rng.choice([1, 2, 7, 30, 31, 32, 33, 63, 64, 65, 90, 127, 128, 129, 200]))
        nq = int(rng.choice([1, 3, 257, 4000]))
        n_ref = int(rng.choice([max(k, 1), 500, 70000]))
        kern = str(rng.choice(["gaussian", "scarches", "inverse_distance", "equal"]))
        idx = rng.integers(0, n_ref, (nq, k)).astype(np.int64)
        if rng.random() < 0.6:
            idx[rng.random((nq, k)) < float(rng.choice([0.02, 0.3]))] = -1
        if k > 1 and rng.random() < 0.5:
            rows = rng.random(nq) < 0.3
            idx[rows, k - 1] = idx[rows, 0]
        if rng.random() < 0.2:
            idx[0] = -1                                     # a row without neighbours
        dist = np.sort(rng.random((nq, k)) * 3.0, axis=1) + 0.05
        dist[idx < 0] = np.inf
        why = ""
        t0 = time.perf_counter()
        t_gpu = 0.0
        t_parts = [0.0, 0.0, 0.0]   # mapping_matrix, label_vote, spmm_dense
        try:
            if not np.isfinite(dist).any():
                raise ValueError("skip")
            al0 = h.alloc_stats()
            tg = time.perf_counter()
            indptr, cols, vals, conn, _ = h.mapping_matrix(dist, idx, n_ref, kern, want_conn=True)
            if time.perf_counter() - tg > 1.0:
                al1 = h.alloc_stats()
                print('SLOW mapping_matrix', round(time.perf_counter() - tg, 2), 's; allocator delta:', {k_: round(al1[k_] - al0[k_], 2) for k_ in al1 if isinstance(al1[k_], (int, float)) and al1[k_] != al0.get(k_)}, flush=True)
            t_gpu += time.perf_counter() - tg; t_parts[0] = time.perf_counter() - tg
            w, _ = orc.kernel_weights(dist, idx, kern)
            oip, oc, ov32, ov64 = orc.mapping_matrix(w, dist, idx, n_ref)
            if not (np.array_equal(indptr, oip) and np.array_equal(cols, oc)):
                why += " structure"
            else:
                if not np.allclose(conn, ov64, rtol=1e-14, atol=0): why += " conn"
                if not np.allclose(vals, ov32, rtol=2e-7, atol=0): why += " vals"
                n_cat = int(rng.choice([1, 2, 20, 300]))
                codes = rng.integers(0, n_cat, n_ref).astype(np.int32)
                tg = time.perf_counter()
                code, conf = h.label_vote(indptr, cols, vals, codes, n_cat)
                t_gpu += time.perf_counter() - tg; t_parts[1] = time.perf_counter() - tg
                ocode, oconf = orc.label_vote(oip, oc, ov32, codes, n_cat)
                if not np.allclose(conf, oconf, rtol=1e-5, atol=1e-7) or (code != ocode).mean() > 5e-3: why += " vote"
                g = int(rng.choice([1, 2, 50, 300]))
                B = rng.standard_normal((n_ref, g)).astype(np.float64 if rng.random() < 0.4 else np.float32)
                tg = time.perf_counter()
                out = h.spmm_dense(oip, oc, ov32, B)
                t_gpu += time.perf_counter() - tg; t_parts[2] = time.perf_counter() - tg
                want = orc.spmm_dense(oip, oc, ov32, B)
                if out.dtype != want.dtype or not np.allclose(out, want, rtol=1e-6, atol=1e-7): why += " spmm"
        except ValueError as e:
            if str(e) != "skip":
                why += " " + str(e)[:60]
        ok = why == ""
        print(json.dumps(dict(case=case, k=k, nq=nq, n_ref=n_ref, kernel=kern, ok=ok, failed=why.strip(),
                              ms=round((time.perf_counter() - t0) * 1e3, 1), ms_library=round(t_gpu * 1e3, 1), ms_calls=[round(x * 1e3, 1) for x in t_parts])), flush=True)
        if not ok:
            bad += 1
    print("FAILED" if bad else f"all {a.cases} cases ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
