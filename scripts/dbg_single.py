import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellmapper_amd import _native
d, k, sigma = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rng = np.random.default_rng(166 + d)
n_ref, nq = 260_000, 150_000
cen = rng.standard_normal((40, d)) * 5.0
ref = (cen[rng.integers(0, 40, n_ref)] + sigma * rng.standard_normal((n_ref, d))).astype(np.float32)
qry = (cen[rng.integers(0, 40, nq)] + sigma * rng.standard_normal((nq, d))).astype(np.float32)
h = _native.default_handle()
os.environ["CELLMAP_NO_HINT"] = "1"
for dbg in ("272", "784"):
    os.environ["CELLMAP_DBG"] = dbg; _native.reload_knobs()
    print("DBG", dbg, flush=True)
    dist, idx = h.knn(ref, qry, k)
    print(h.knn_stats(), flush=True)
