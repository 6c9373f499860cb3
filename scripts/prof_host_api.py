import cProfile, pstats, io, json, os, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cellmapper_amd import CellMapper, MiniAnnData
n = 1_500_000
rng = np.random.default_rng(0)
ref = rng.standard_normal((n, 50), dtype=np.float32); qry = rng.standard_normal((n, 50), dtype=np.float32)
labels = pd.Categorical.from_codes(rng.integers(0, 20, n), categories=[f"c{i:02d}" for i in range(20)])
R = MiniAnnData(X=np.zeros((n, 1), np.float32), obs=pd.DataFrame({"label": labels}), obsm={"X_pca": ref})
Q = MiniAnnData(X=np.zeros((n, 1), np.float32), obsm={"X_pca": qry})
os.environ["LOGLEVEL"] = "WARNING"
for rep in range(3):
    cm = CellMapper(Q, R)
    if rep == 2:
        pr0 = cProfile.Profile(); pr0.enable()
    t0 = time.perf_counter(); cm.compute_neighbors(n_neighbors=30, use_rep="X_pca", only_yx=True); t1 = time.perf_counter()
    if rep == 2:
        pr0.disable(); s0 = io.StringIO(); pstats.Stats(pr0, stream=s0).sort_stats("tottime").print_stats(14); print(s0.getvalue()[:3000])
    cm.compute_mapping_matrix("gaussian"); t2 = time.perf_counter()
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    cm.map_obs("label"); t3 = time.perf_counter()
    if rep == 2:
        pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(6); print(s.getvalue()[:1500])
    print(json.dumps({"knn_s": t1 - t0, "mapping_matrix_s": t2 - t1, "map_obs_s": t3 - t2, "total_s": t3 - t0}))
