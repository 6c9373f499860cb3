#!/usr/bin/env python3
"""Where does the host-in -> host-out time of config 2 go?  Times every call of the graph-tier flow that
`CellMapper.map(obs_keys=...)` makes (cm_rep_create x2, cm_graph_knn, cm_graph_mapping, cm_graph_label_vote, destroys)
over several iterations on one handle; CELLMAP_XFER=1 gives the staging-ring arm of the A/B (default: plain hipMemcpy)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from cellmapper_amd import _native
    n = int(os.environ.get("HL_N", 1_500_000)); d, k, ncat = 50, 30, 20
    iters = int(os.environ.get("HL_ITERS", 6))
    rng = np.random.default_rng(0)
    ref = rng.standard_normal((n, d), dtype=np.float32); qry = rng.standard_normal((n, d), dtype=np.float32)
    codes = rng.integers(0, ncat, n).astype(np.int32)
    h = _native.Handle(0)
    rows = []
    for it in range(iters):
        t = [time.perf_counter()]
        xr = _native.DeviceRep(h, ref); t.append(time.perf_counter())
        yr = _native.DeviceRep(h, qry); t.append(time.perf_counter())
        g = _native.DeviceGraph(h, xr, yr, k, "euclidean"); t.append(time.perf_counter())
        xr.close(); yr.close(); t.append(time.perf_counter())
        g.mapping("gaussian"); t.append(time.perf_counter())
        code, conf = g.label_vote(codes, ncat); t.append(time.perf_counter())
        g.close(); t.append(time.perf_counter())
        names = ["rep_ref", "rep_qry", "graph_knn", "rep_close", "mapping", "label_vote", "graph_close"]
        rows.append({nm: (t[i + 1] - t[i]) * 1e3 for i, nm in enumerate(names)} | {"total_to_results": (t[6] - t[0]) * 1e3})
    out = {"xfer": os.environ.get("CELLMAP_XFER", "off"), "n": n, "iterations_ms": rows}
    if os.environ.get("HL_FETCH"):
        xr = _native.DeviceRep(h, ref); yr = _native.DeviceRep(h, qry)
        g = _native.DeviceGraph(h, xr, yr, k, "euclidean")
        for _ in range(2):
            t0 = time.perf_counter(); dist, idx = g.fetch_knn(); out.setdefault("fetch_knn_ms", []).append((time.perf_counter() - t0) * 1e3)
        g.mapping("gaussian")
        t0 = time.perf_counter(); m = g.fetch_mapping(); out["fetch_mapping_ms"] = (time.perf_counter() - t0) * 1e3
        out["fetch_checksum"] = [float(dist.sum()), int(idx.sum() % 1000003)]
    print(json.dumps(out))


main()
