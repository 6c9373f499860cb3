#!/bin/bash
# the direct-difference VALU kernel at d=50 (candidate for a high-precision filter level): parity on a small case + time
python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
import torch
from cellmapper_amd import _native
rng = np.random.default_rng(0)
ref = rng.standard_normal((20000, 50)).astype(np.float32); qry = rng.standard_normal((3000, 50)).astype(np.float32)
h = _native.default_handle()
d0, i0 = h.knn(ref, qry, 30)
os.environ["CELLMAP_PHASE1"] = "direct"
d1, i1 = h.knn(ref, qry, 30)
print("direct kernel parity:", np.array_equal(i0, i1), np.array_equal(d0, d1), h.knn_stats())
PY
CELLMAP_PHASE1=direct timeout -k 10 200 python scripts/bench_knn.py --nq 750000 --nref 1500000 --d 50 --k 30 --uniform 0 --reps 1 2>&1 | tail -1 | cut -c1-400
