#!/usr/bin/env python3
"""k-NN stage lines of the bench line's ragged leg (bench_data.mixture_rows(..., ragged=True), sigma 0.5, 1.5 M x 1.5 M, d = 50):
run with CELLMAP_DBG=16."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_data as bd
from cellmapper_amd.device import DevicePipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ragged = (sys.argv[3] != "equal") if len(sys.argv) > 3 else True
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
pipe = DevicePipeline(0); pipe.h.set_profiling(True); torch.cuda.set_stream(pipe.stream)
dev = pipe.device
ref = bd.mixture_rows(torch, bd.STREAM_REF, 0, n, 50, dev, 40, 5.0, sigma, ragged=ragged)
qry = bd.mixture_rows(torch, bd.STREAM_QRY, 0, n, 50, dev, 40, 5.0, sigma, ragged=ragged)
for _ in range(reps):
    dist, idx = pipe.knn(ref, qry, 30)
    torch.cuda.synchronize()
    print(json.dumps({k: round(v, 1) for k, v in pipe.h.stage_times().items() if v > 0.05}), file=sys.stderr)
torch.cuda.synchronize()
st = pipe.h.stage_times(); ks = pipe.h.knn_stats()
print(json.dumps({"stage_ms": {k: round(v, 1) for k, v in st.items() if v > 0.05}, "stats": ks}))
