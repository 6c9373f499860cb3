#!/usr/bin/env python3
"""End-to-end timing of the drop-in host API (numpy in / numpy out, PCIe copies included) at BASELINE config 2."""
import json, os, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellmapper_amd import CellMapper, MiniAnnData
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
rng = np.random.default_rng(0)
ref = rng.standard_normal((n, 50), dtype=np.float32); qry = rng.standard_normal((n, 50), dtype=np.float32)
labels = pd.Categorical.from_codes(rng.integers(0, 20, n), categories=[f"c{i:02d}" for i in range(20)])
R = MiniAnnData(X=np.zeros((n, 1), np.float32), obs=pd.DataFrame({"label": labels}), obsm={"X_pca": ref})
Q = MiniAnnData(X=np.zeros((n, 1), np.float32), obsm={"X_pca": qry})
os.environ["LOGLEVEL"] = "WARNING"
out = {}
for rep in range(2):
    cm = CellMapper(Q, R)
    t0 = time.perf_counter(); cm.compute_neighbors(n_neighbors=30, use_rep="X_pca", only_yx=True); t1 = time.perf_counter()
    cm.compute_mapping_matrix("gaussian"); t2 = time.perf_counter()
    cm.map_obs("label"); t3 = time.perf_counter()
    out = {"n": n, "knn_s": t1 - t0, "mapping_matrix_s": t2 - t1, "map_obs_s": t3 - t2, "total_s": t3 - t0, "cells_per_s": n / (t3 - t0)}
print(json.dumps(out))
