#!/usr/bin/env python3
"""What does the FIRST call of a problem pay that the following ones do not?  Config 2's fused device-tier call
(cm_map_labels_dev) on a fresh handle: wall time and HIP-event stage times of calls 1, 2, 3 and the mean of the next ten;
run it with CELLMAP_NO_HINT=1 to see every call take the un-hinted route (pilot, pruning gate, locality sort)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench_data as bd
    from cellmapper_amd.device import DevicePipeline

    nq = int(os.environ.get("FC_NQ", 1_500_000)); nref = int(os.environ.get("FC_NREF", 1_500_000)); d, k, ncat = 50, 30, 20
    dev = torch.device("cuda", 0)
    ref = bd.fill_rows(torch, "normal", bd.STREAM_REF, 0, nref, d, dev)
    qry = bd.fill_rows(torch, "normal", bd.STREAM_QRY, 0, nq, d, dev)
    codes = bd.integers(torch, bd.STREAM_LABEL, 0, nref, ncat, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe = DevicePipeline(0)
    t_create = (time.perf_counter() - t0) * 1e3
    pipe.h.set_profiling(True)
    torch.cuda.set_stream(pipe.stream)
    rows = []
    for i in range(13):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.map_labels(ref, qry, k, codes, ncat)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        st = pipe.h.stage_times()
        rows.append((wall, st))
    out = {"no_hint": bool(os.environ.get("CELLMAP_NO_HINT")), "handle_create_ms": t_create,
           "call1": {"wall_ms": rows[0][0], "stages": rows[0][1]}, "call2": {"wall_ms": rows[1][0], "stages": rows[1][1]},
           "call3": {"wall_ms": rows[2][0], "stages": rows[2][1]},
           "mean_calls_4_13_wall_ms": sum(r[0] for r in rows[3:]) / 10.0,
           "mean_calls_4_13_stages": {kk: sum(r[1][kk] for r in rows[3:]) / 10.0 for kk in rows[3][1]}}
    print(json.dumps(out))


main()
