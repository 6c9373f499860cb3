"""ctypes binding of ``libcellmap_hip.so`` (C ABI: ``include/cellmap.h``).

This is the only place the Python layer touches native code.  There is deliberately NO CPU fallback: if the
shared library or a GPU is missing, every entry point raises ``RuntimeError`` with a build hint, mirroring
the reference's optional-backend gate (``check.py:39-56``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CELLMAP_HIP_LIB", os.path.join(_HERE, "libcellmap_hip.so"))

BUILD_HINT = (
    "The 'hip' backend needs libcellmap_hip.so (gfx950 / MI355X). Build it in-tree with "
    "`make -C cellmapper_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`) on a ROCm >= 7.0 box."
)

CM_F32, CM_F64 = 0, 1
# metric names as sklearn spells them (the reference hands `metric` to sklearn, knn.py:343,429); minkowski: p = 2, its default
METRICS = {"euclidean": 0, "l2": 0, "minkowski": 0, "cosine": 1, "manhattan": 2, "cityblock": 2, "l1": 2, "chebyshev": 3, "infinity": 3, "sqeuclidean": 4, "correlation": 5,
           "canberra": 6, "braycurtis": 7,
           # the counting metrics sklearn takes without parameters (scipy cdist on boolean copies / DistanceMetric on a ball tree)
           "hamming": 8, "jaccard": 9, "dice": 10, "rogerstanimoto": 11, "russellrao": 12, "sokalmichener": 13, "sokalsneath": 14, "yule": 15,
           # round 5: (latitude, longitude) rows in radians; NaN = missing coordinate
           "haversine": 16, "nan_euclidean": 17}
CM_REP_ALLOW_NAN = 1
SUPPORTED_METRICS_TEXT = ("euclidean / l2 / minkowski (p = 2), sqeuclidean, cosine, correlation, manhattan / cityblock / l1, chebyshev / infinity, "
                          "canberra, braycurtis, hamming, jaccard, dice, rogerstanimoto, russellrao, sokalmichener, sokalsneath, yule, haversine, nan_euclidean")
FORMS = {"auto": 0, "gemm": 1, "direct": 2}
KERNELS = {"gaussian": 0, "scarches": 1, "inverse_distance": 2, "equal": 3, "_identity": 4}
CM_KNN_SQUARED, CM_KNN_FORCE_FALLBACK, CM_KNN_F64_OUT = 1, 2, 4

_STATUS_EXC = {1: ValueError, 2: MemoryError, 3: RuntimeError, 4: ValueError, 5: NotImplementedError, 6: RuntimeError}

_lib = None
_lib_lock = threading.Lock()

EXPORTS = [
    "cm_version", "cm_device_count", "cm_create", "cm_create_multi", "cm_destroy", "cm_pool_trim", "cm_workspace_bytes", "cm_workspace_bytes_all_routes", "cm_reserve_workspace", "cm_alloc_stats", "cm_reload_knobs", "cm_last_error", "cm_set_stream", "cm_synchronize",
    "cm_knn_stats", "cm_knn", "cm_knn_dev", "cm_knn_merge_dev", "cm_knn_merge_strided_dev", "cm_mapping_matrix", "cm_dist_moments_dev",
    "cm_kernel_param_dev", "cm_weights_dev", "cm_ell_to_csr_dev", "cm_label_vote", "cm_label_vote_dev",
    "cm_spmm_dense", "cm_spmm_dense_dev", "cm_spmm_csr_nnz", "cm_spmm_csr", "cm_spmm_csr_nnz_dev", "cm_spmm_csr_dev", "cm_spmm_csr_fused_dev", "cm_payload_index_len", "cm_payload_index_dev", "cm_locality_order_dev", "cm_map_labels_dev", "cm_csr_normalize",
    "cm_set_profiling", "cm_stage_times", "cm_shared_neighbors", "cm_presence_scores", "cm_debug_mfma_f16", "cm_debug_mfma_f16_k32", "cm_graph_part_info",
    "cm_ingest_distances", "cm_expression_metrics", "cm_expression_metrics_dev",
    "cm_rep_create", "cm_rep_create_ex", "cm_rep_destroy", "cm_graph_knn", "cm_graph_destroy", "cm_graph_info", "cm_graph_knn_fetch",
    "cm_graph_mapping", "cm_graph_mapping_fetch", "cm_graph_label_vote", "cm_graph_spmm_dense", "cm_graph_spmm_csr_nnz",
    "cm_graph_spmm_csr", "cm_gram", "cm_project",
]


def load_library():
    """dlopen the shared library and declare argument types (no GPU call is made here)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found. {BUILD_HINT}")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # missing ROCm runtime etc.
            raise RuntimeError(f"Could not load {LIB_PATH}: {e}. {BUILD_HINT}") from e
        vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
        lib.cm_version.restype = C.c_int
        lib.cm_device_count.restype = C.c_int
        lib.cm_create.argtypes = [C.c_int, C.POINTER(vp)]
        lib.cm_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        lib.cm_destroy.argtypes = [vp]
        lib.cm_pool_trim.argtypes = [vp]
        lib.cm_workspace_bytes.argtypes = [i64, i64, i32, i32, i32, i32]
        lib.cm_workspace_bytes.restype = i64
        lib.cm_workspace_bytes_all_routes.argtypes = [i64, i64, i32, i32, i32, i32]
        lib.cm_workspace_bytes_all_routes.restype = i64
        lib.cm_reserve_workspace.argtypes = [vp, i64]
        lib.cm_alloc_stats.argtypes = [vp, vp]
        lib.cm_reload_knobs.argtypes = []
        lib.cm_last_error.argtypes = [vp]
        lib.cm_last_error.restype = C.c_char_p
        lib.cm_set_stream.argtypes = [vp, vp]
        lib.cm_synchronize.argtypes = [vp]
        lib.cm_knn_stats.argtypes = [vp, vp]
        lib.cm_knn.argtypes = [vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp, vp]
        lib.cm_knn_dev.argtypes = [vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, i64, i32, vp, vp]
        lib.cm_knn_merge_dev.argtypes = [vp, vp, vp, i32, i64, i32, i32, vp, vp]
        lib.cm_knn_merge_strided_dev.argtypes = [vp, vp, vp, i32, i64, i64, i32, i32, vp, vp]
        lib.cm_mapping_matrix.argtypes = [vp, vp, vp, i64, i32, i64, i32, dbl, vp, vp, vp, vp, vp, vp]
        lib.cm_dist_moments_dev.argtypes = [vp, vp, vp, i64, vp, vp]
        lib.cm_kernel_param_dev.argtypes = [vp, vp, i32, vp]
        lib.cm_weights_dev.argtypes = [vp, vp, vp, i64, i32, i64, i32, vp, dbl, vp, vp, vp, vp]
        lib.cm_ell_to_csr_dev.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp, vp]
        lib.cm_label_vote.argtypes = [vp, vp, vp, vp, i64, vp, i64, i32, vp, vp]
        lib.cm_label_vote_dev.argtypes = [vp, vp, vp, vp, i64, vp, i64, i32, vp, vp]
        lib.cm_spmm_dense.argtypes = [vp, vp, vp, vp, i64, vp, i64, i64, i32, vp]
        lib.cm_spmm_dense_dev.argtypes = [vp, vp, vp, vp, i64, vp, i64, i64, i32, vp]
        lib.cm_spmm_csr_nnz.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, vp]
        lib.cm_spmm_csr.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, vp, vp, vp]
        lib.cm_spmm_csr_nnz_dev.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, vp, vp]
        lib.cm_spmm_csr_dev.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, vp, vp, vp, vp]
        lib.cm_spmm_csr_fused_dev.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, vp, vp]
        lib.cm_payload_index_len.argtypes = [i64, i64]
        lib.cm_payload_index_len.restype = i64
        lib.cm_payload_index_dev.argtypes = [vp, vp, vp, i64, i64, vp]
        lib.cm_locality_order_dev.argtypes = [vp, vp, vp, i64, i64, vp]
        lib.cm_map_labels_dev.argtypes = [vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, dbl, vp, i32, vp, vp, vp, vp]
        lib.cm_csr_normalize.argtypes = [vp, vp, vp, i64, vp]
        lib.cm_shared_neighbors.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, vp, vp, i64, vp]
        lib.cm_presence_scores.argtypes = [vp, vp, vp, i64, i32, i64, vp, i32, vp]
        lib.cm_set_profiling.argtypes = [vp, i32]
        lib.cm_stage_times.argtypes = [vp, vp]
        lib.cm_ingest_distances.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp, vp]
        lib.cm_expression_metrics.argtypes = [vp, vp, vp, i64, i64, vp, i32, i32, vp]
        lib.cm_expression_metrics_dev.argtypes = [vp, vp, vp, i64, i64, vp, i32, i32, vp]
        lib.cm_debug_mfma_f16.argtypes = [vp, vp, vp, vp, vp, i32, i32]
        lib.cm_debug_mfma_f16_k32.argtypes = [vp, vp, vp, vp, vp, i32, i32]
        lib.cm_rep_create.argtypes = [vp, vp, i64, i32, i32, C.POINTER(vp)]
        lib.cm_rep_create_ex.argtypes = [vp, vp, i64, i32, i32, i32, C.POINTER(vp)]
        lib.cm_rep_destroy.argtypes = [vp, vp]
        lib.cm_graph_knn.argtypes = [vp, vp, vp, i32, i32, i32, C.POINTER(vp)]
        lib.cm_graph_destroy.argtypes = [vp, vp]
        lib.cm_graph_info.argtypes = [vp, vp, vp]
        lib.cm_graph_part_info.argtypes = [vp, vp, i32, vp]
        lib.cm_graph_knn_fetch.argtypes = [vp, vp, vp, vp]
        lib.cm_graph_mapping.argtypes = [vp, vp, i32, dbl, vp, vp]
        lib.cm_graph_mapping_fetch.argtypes = [vp, vp, vp, vp, vp]
        lib.cm_graph_label_vote.argtypes = [vp, vp, vp, i64, i32, vp, vp]
        lib.cm_graph_spmm_dense.argtypes = [vp, vp, vp, i64, i64, i32, vp]
        lib.cm_graph_spmm_csr_nnz.argtypes = [vp, vp, vp, vp, vp, i64, i64, vp]
        lib.cm_graph_spmm_csr.argtypes = [vp, vp, vp, vp, vp, i64, i64, vp, vp, vp]
        lib.cm_gram.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
        lib.cm_project.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, i32, vp]
        lib.cm_version.argtypes = []
        lib.cm_device_count.argtypes = []
        for name in EXPORTS:
            fn = getattr(lib, name)
            if name not in ("cm_version", "cm_device_count", "cm_last_error", "cm_payload_index_len", "cm_workspace_bytes", "cm_workspace_bytes_all_routes"):
                fn.restype = C.c_int
        _lib = lib
        return lib


def reload_knobs():
    """Re-read the ``CELLMAP_*`` diagnostics / test variables (the library reads them in ``cm_create`` only)."""
    load_library().cm_reload_knobs()


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))  # raw device pointer (e.g. torch.Tensor.data_ptr())


class Handle:
    """One backend context (``cm_create`` / ``cm_create_multi`` / ``cm_destroy``): one GPU + stream + workspace, or -- with
    ``devices=[...]`` -- several GPUs of one node behind ONE handle: the graph tier (what the drop-in classes call) then
    shards the reference rows over them, exchanges the per-shard lists with peer copies and merges exactly, all inside the
    library (one host thread per device; ctypes releases the GIL)."""

    def __init__(self, device: int = 0, devices=None):
        self.lib = load_library()
        h = C.c_void_p()
        if devices is not None:
            devices = [int(x) for x in devices]
            if not devices:
                raise ValueError("Handle(devices=...): empty device list")
            arr = (C.c_int * len(devices))(*devices)
            rc = self.lib.cm_create_multi(arr, len(devices), C.byref(h))
            device = devices[0]
        else:
            rc = self.lib.cm_create(int(device), C.byref(h))
            devices = [int(device)]
        if rc != 0:
            msg = self.lib.cm_last_error(None).decode()
            raise RuntimeError(f"Could not initialise the HIP backend on device(s) {devices}: {msg}. {BUILD_HINT}")
        self._h = h
        self.device = device
        self.devices = devices

    def close(self):
        if getattr(self, "_h", None):
            self.lib.cm_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != 0:
            msg = self.lib.cm_last_error(self._h).decode()
            raise _STATUS_EXC.get(rc, RuntimeError)(msg or f"cellmap status {rc}")

    def call(self, name: str, *args):
        self.check(getattr(self.lib, name)(self._h, *args))

    def trim(self):
        """Return the handle's cached device blocks to the HIP allocator (``cm_pool_trim``)."""
        self.call("cm_pool_trim")

    def reserve(self, n_ref: int, nq: int, d: int, k: int, dtype=np.float32, metric: str = "euclidean", all_routes: bool = False):
        """Allocate the device workspace of a problem of this shape NOW (``cm_reserve_workspace``): the first search of that
        problem then allocates nothing.  ``all_routes``: also what clustered embeddings add on demand (bucket pruning, the
        bucket-centred level).  Returns the bytes reserved."""
        fn = self.lib.cm_workspace_bytes_all_routes if all_routes else self.lib.cm_workspace_bytes
        b = int(fn(int(n_ref), int(nq), int(d), CM_F64 if np.dtype(dtype) == np.float64 else CM_F32, int(k), METRICS[metric]))
        if b < 0:
            raise ValueError("reserve: bad problem shape")
        self.call("cm_reserve_workspace", b)
        return b

    def alloc_stats(self):
        """Wall time this handle has spent inside the device allocator since it was created (``cm_alloc_stats``)."""
        o = np.zeros(16, np.float64)
        self.call("cm_alloc_stats", _ptr(o))
        kinds = ["workspace", "pool", "free", "pinned"]
        rec = {f"{kinds[i]}_ms": float(o[i]) for i in range(4)}
        rec.update({f"{kinds[i]}_bytes": float(o[4 + i]) for i in range(4) if i != 2})
        rec.update({f"{kinds[i]}_calls": int(o[8 + i]) for i in range(4)})
        rec.update(create_ms=float(o[12]), preload_ms=float(o[13]), workspace_capacity=float(o[14]), workspace_peak_bytes=float(o[15]))
        return rec

    def set_stream(self, stream_ptr):
        self.call("cm_set_stream", C.c_void_p(int(stream_ptr) if stream_ptr else 0))

    def synchronize(self):
        self.call("cm_synchronize")

    def set_profiling(self, on: bool):
        self.call("cm_set_profiling", int(bool(on)))

    def stage_times(self):
        """Milliseconds per stage of the last device-tier call (HIP events on the handle's stream)."""
        out = np.zeros(8, np.float32)
        self.call("cm_stage_times", _ptr(out))
        names = ["", "prep", "phase1_mfma", "rerank", "fallback", "stats", "weights", "vote"]
        return {names[i]: float(out[i]) for i in range(1, 8)}

    def knn_stats(self):
        out = np.zeros(8, np.int64)
        self.call("cm_knn_stats", _ptr(out))
        # out[3]: 1 = f32-input MFMA, 2 = split-fp16 MFMA (K<=64), 3 = low-d VALU, 4 = split-fp16 MFMA (K=128)
        return dict(fallback_queries=int(out[0]), candidates=int(out[1]), ref_splits=int(out[2]) % 100,
                    filter_products=(int(out[2]) // 100) % 100, filter_kernel=int(out[2]) // 10000, lowd_path=int(out[3]) % 100,
                    # bucket pruning: (query block, reference tile) pairs visited, per cent of a full scan (0 = not used)
                    tiles_visited_pct=(int(out[3]) // 100) % 1000, pruning_passes=int(out[3]) // 100000,
                    # 1 = a query's exact-fallback list overflowed (results truncated): device-tier callers must check
                    fallback_overflow=int(out[4]),
                    # most reference splits a subset stage used (> 16: a small seeded subset split to fill the chip)
                    subset_splits=int(out[5]), wide_split_retries=int(out[6]))

    # ------------------------------------------------------------------ host tier (numpy in / numpy out)
    def knn(self, ref: np.ndarray, qry: np.ndarray, k: int, metric: str = "euclidean", form: str = "auto"):
        if metric not in METRICS:
            raise ValueError(f"Metric '{metric}' is not supported by the hip backend (supported: {SUPPORTED_METRICS_TEXT}).")
        ref = np.ascontiguousarray(ref)
        qry = np.ascontiguousarray(qry)
        if ref.ndim != 2 or qry.ndim != 2 or ref.shape[1] != qry.shape[1]:
            raise ValueError(f"Incompatible representation shapes {ref.shape} and {qry.shape}.")
        dt = np.float32 if (ref.dtype == np.float32 and qry.dtype == np.float32) else np.float64
        ref = ref.astype(dt, copy=False)
        qry = qry.astype(dt, copy=False)
        # (NaN / infinity: screened by the library on the device copies -> CM_E_INVALID -> ValueError, like sklearn's check_array)
        n_ref, d = ref.shape
        nq = qry.shape[0]
        k = int(k)
        out_dt = dt if METRICS[metric] == 1 else np.float64
        dist = np.empty((nq, k), out_dt)
        idx = np.empty((nq, k), np.int64)
        self.call("cm_knn", _ptr(ref), n_ref, _ptr(qry), nq, d, CM_F64 if dt == np.float64 else CM_F32, k,
                  METRICS[metric], FORMS[form], _ptr(dist), _ptr(idx))
        if metric == "nan_euclidean" and dt == np.float32:
            dist = dist.astype(np.float32)   # (sklearn returns float32 for float32 data; the library's float64 values are float32-rounded already)
        return dist, idx

    def mapping_matrix(self, dist, idx, n_ref: int, kernel: str = "gaussian", epsilon: float = 1e-8, want_conn=False):
        """-> (indptr int64, cols int32, vals float32 normalised, conn float64 | None, param)."""
        if kernel not in KERNELS:
            raise ValueError(
                f"Unknown kernel: {kernel}. Supported kernels are: 'gaussian', 'scarches', 'random', "
                "'inverse_distance', 'equal'.")
        dist = np.ascontiguousarray(dist, np.float64)
        idx = np.ascontiguousarray(idx, np.int64)
        if dist.shape != idx.shape:
            raise ValueError("Indices and distances must have the same shape.")
        nq, k = idx.shape
        if idx.size and (idx.max() >= n_ref or idx.min() < -1):
            raise ValueError("column index exceeds matrix dimensions")
        indptr = np.empty(nq + 1, np.int64)
        cols = np.empty(nq * k, np.int32)
        vals = np.empty(nq * k, np.float32)
        conn = np.empty(nq * k, np.float64) if want_conn else None
        nnz = C.c_int64(0)
        param = C.c_double(0.0)
        self.call("cm_mapping_matrix", _ptr(dist), _ptr(idx), nq, k, int(n_ref), KERNELS[kernel], float(epsilon),
                  _ptr(indptr), _ptr(cols), _ptr(vals), _ptr(conn), C.byref(nnz), C.byref(param))
        n = nnz.value
        return indptr, cols[:n], vals[:n], (conn[:n] if want_conn else None), param.value

    def csr_normalize(self, indptr, data64):
        indptr = np.ascontiguousarray(indptr, np.int64)
        data64 = np.ascontiguousarray(data64, np.float64)
        out = np.empty(data64.shape[0], np.float32)
        self.call("cm_csr_normalize", _ptr(indptr), _ptr(data64), len(indptr) - 1, _ptr(out))
        return out

    def shared_neighbors(self, yx, xx, yy, xy, transform: str = "jaccard"):
        """Shared-neighbour matrix of the four k-NN graphs -> (indptr int64, cols int32, vals float64), canonical CSR."""
        tid = {"counts": 0, "jaccard": 1, "hnoca": 2}[transform]
        yx, xx, yy, xy = (np.ascontiguousarray(a, np.int64) for a in (yx, xx, yy, xy))
        nq, k = yx.shape
        nr = xx.shape[0]
        if yy.shape != (nq, k) or xx.shape != (nr, k) or xy.shape != (nr, k):
            raise ValueError("The four neighbor graphs must share n_neighbors and have consistent shapes.")
        indptr = np.empty(nq + 1, np.int64)
        nnz = C.c_int64(0)
        self.call("cm_shared_neighbors", _ptr(yx), _ptr(xx), _ptr(yy), _ptr(xy), nq, nr, k, tid, _ptr(indptr), None, None,
                  0, C.byref(nnz))
        cols = np.empty(nnz.value, np.int32)
        vals = np.empty(nnz.value, np.float64)
        self.call("cm_shared_neighbors", _ptr(yx), _ptr(xx), _ptr(yy), _ptr(xy), nq, nr, k, tid, _ptr(indptr), _ptr(cols),
                  _ptr(vals), nnz.value, C.byref(nnz))
        return indptr, cols, vals

    def ingest_distances(self, indptr, indices, data, n: int, include_self):
        """Square CSR distance matrix -> padded (indices int64, distances float64) [n, kmax] (utils.py:129-219)."""
        indptr = np.ascontiguousarray(indptr, np.int64)
        indices = np.ascontiguousarray(indices, np.int32)
        data = np.ascontiguousarray(data, np.float64)
        mode = -1 if include_self is None else (1 if include_self else 0)
        km = C.c_int32(0)
        self.call("cm_ingest_distances", _ptr(indptr), _ptr(indices), _ptr(data), int(n), mode, 0, None, None, C.byref(km))
        idx = np.empty((int(n), km.value), np.int64)
        dist = np.empty((int(n), km.value), np.float64)
        if idx.size:
            self.call("cm_ingest_distances", _ptr(indptr), _ptr(indices), _ptr(data), int(n), mode, km.value, _ptr(idx),
                      _ptr(dist), C.byref(km))
        return idx, dist

    def expression_metrics(self, orig, imputed, method: str, group=None, n_groups: int = 0):
        """Per-gene agreement [n_groups + 1, G] float32 (row 0: all cells) -- evaluate.py:236-323."""
        mid = {"pearson": 0, "rmse": 1, "js": 2, "jensen-shannon": 2, "spearman": 3}[method]
        orig = np.ascontiguousarray(orig, np.float32)
        imputed = np.ascontiguousarray(imputed, np.float32)
        if orig.shape != imputed.shape or orig.ndim != 2:
            raise ValueError(f"Expression arrays must share a 2-D shape (got {orig.shape} and {imputed.shape}).")
        grp = None if group is None else np.ascontiguousarray(group, np.int32)
        ng = int(n_groups) if grp is not None else 0
        out = np.empty((ng + 1, orig.shape[1]), np.float32)
        self.call("cm_expression_metrics", _ptr(orig), _ptr(imputed), orig.shape[0], orig.shape[1], _ptr(grp), ng, mid,
                  _ptr(out))
        return out

    def presence_scores(self, dist, idx, n_ref: int, group=None, n_groups: int = 0):
        """Raw presence scores [n_ref, n_groups + 1] (column 0: all query cells)."""
        dist = np.ascontiguousarray(dist, np.float64)
        idx = np.ascontiguousarray(idx, np.int64)
        nq, k = idx.shape
        grp = None if group is None else np.ascontiguousarray(group, np.int32)
        out = np.empty((int(n_ref), int(n_groups) + 1), np.float64)
        self.call("cm_presence_scores", _ptr(dist), _ptr(idx), nq, k, int(n_ref), _ptr(grp), int(n_groups), _ptr(out))
        return out

    # ---- joint representations: float64 Gram matrix / projection of a cell x gene matrix
    @staticmethod
    def _tall(X):
        """(dense, indptr, cols, vals, n, p, dtype_id) of a dense or scipy-sparse cell x gene matrix."""
        import scipy.sparse as sp

        if sp.issparse(X):
            X = X.tocsr()
            if not X.has_canonical_format:
                X = X.copy()
                X.sum_duplicates()
            return (None, np.ascontiguousarray(X.indptr, np.int64), np.ascontiguousarray(X.indices, np.int32),
                    np.ascontiguousarray(X.data, np.float32), X.shape[0], X.shape[1], CM_F32)
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        return X, None, None, None, X.shape[0], X.shape[1], CM_F64 if X.dtype == np.float64 else CM_F32

    def gram(self, X, center_rows: bool = False):
        """``A'^T A'`` [p, p] float64 (``A'`` = rows centred at their means if ``center_rows``)."""
        D, ip, ic, iv, n, p, dt = self._tall(X)
        out = np.empty((p, p), np.float64)
        self.call("cm_gram", _ptr(D), _ptr(ip), _ptr(ic), _ptr(iv), n, p, dt, int(bool(center_rows)), _ptr(out))
        return out

    def project(self, X, P, center_rows: bool = False, col_shift=None):
        """``A' P`` [n, nc] float64; ``A'`` = A, rows centred (``center_rows``) or shifted by ``col_shift`` [p]."""
        D, ip, ic, iv, n, p, dt = self._tall(X)
        P = np.ascontiguousarray(P, np.float64)
        if P.ndim != 2 or P.shape[0] != p:
            raise ValueError(f"Projection matrix must have shape ({p}, nc); got {P.shape}.")
        mode = 1 if center_rows else (2 if col_shift is not None else 0)
        sh = None if col_shift is None else np.ascontiguousarray(col_shift, np.float64)
        out = np.empty((n, P.shape[1]), np.float64)
        for c0 in range(0, P.shape[1], 64):   # the kernel projects onto at most 64 directions per pass
            Pc = np.ascontiguousarray(P[:, c0:c0 + 64])
            oc = np.empty((n, Pc.shape[1]), np.float64)
            self.call("cm_project", _ptr(D), _ptr(ip), _ptr(ic), _ptr(iv), n, p, dt, mode, _ptr(sh), _ptr(Pc), Pc.shape[1], _ptr(oc))
            out[:, c0:c0 + 64] = oc
        return out

    def label_vote(self, indptr, cols, vals, codes, n_cat: int):
        indptr = np.ascontiguousarray(indptr, np.int64)
        cols = np.ascontiguousarray(cols, np.int32)
        vals = np.ascontiguousarray(vals, np.float32)
        codes = np.ascontiguousarray(codes, np.int32)
        nq = len(indptr) - 1
        out_code = np.empty(nq, np.int32)
        out_conf = np.empty(nq, np.float32)
        self.call("cm_label_vote", _ptr(indptr), _ptr(cols), _ptr(vals), nq, _ptr(codes), len(codes), int(n_cat),
                  _ptr(out_code), _ptr(out_conf))
        return out_code, out_conf

    def spmm_dense(self, indptr, cols, vals, B):
        indptr = np.ascontiguousarray(indptr, np.int64)
        cols = np.ascontiguousarray(cols, np.int32)
        vals = np.ascontiguousarray(vals, np.float32)
        B = np.asarray(B)
        squeeze = B.ndim == 1
        if squeeze:
            B = B.reshape(-1, 1)
        if B.dtype != np.float32:
            B = B.astype(np.float64, copy=False)  # scipy upcast(float32, other) -> float64 for ints / float64
        B = np.ascontiguousarray(B)
        nq = len(indptr) - 1
        out = np.empty((nq, B.shape[1]), B.dtype)
        self.call("cm_spmm_dense", _ptr(indptr), _ptr(cols), _ptr(vals), nq, _ptr(B), B.shape[0], B.shape[1],
                  CM_F64 if B.dtype == np.float64 else CM_F32, _ptr(out))
        return out[:, 0] if squeeze else out

    def spmm_csr(self, indptr, cols, vals, B_indptr, B_cols, B_vals, n_out: int):
        indptr = np.ascontiguousarray(indptr, np.int64)
        cols = np.ascontiguousarray(cols, np.int32)
        vals = np.ascontiguousarray(vals, np.float32)
        bi = np.ascontiguousarray(B_indptr, np.int64)
        bc = np.ascontiguousarray(B_cols, np.int32)
        bv = np.ascontiguousarray(B_vals, np.float32)
        nq, n_ref = len(indptr) - 1, len(bi) - 1
        out_indptr = np.empty(nq + 1, np.int64)
        self.call("cm_spmm_csr_nnz", _ptr(indptr), _ptr(cols), _ptr(vals), nq, _ptr(bi), _ptr(bc), _ptr(bv), n_ref,
                  int(n_out), _ptr(out_indptr))
        nnz = int(out_indptr[nq])
        out_cols = np.empty(nnz, np.int32)
        out_vals = np.empty(nnz, np.float32)
        self.call("cm_spmm_csr", _ptr(indptr), _ptr(cols), _ptr(vals), nq, _ptr(bi), _ptr(bc), _ptr(bv), n_ref,
                  int(n_out), _ptr(out_indptr), _ptr(out_cols), _ptr(out_vals))
        return out_indptr, out_cols, out_vals


class DeviceRep:
    """An embedding matrix uploaded once (``cm_rep_create``): ``xrep`` / ``yrep`` of ``Neighbors``."""

    def __init__(self, handle: Handle, X: np.ndarray, allow_nan: bool = False):
        """``allow_nan``: NaN entries are kept as missing coordinates (``cm_rep_create_ex`` with CM_REP_ALLOW_NAN) -- what sklearn
        allows under ``metric="nan_euclidean"`` only; any other metric over such a representation raises like a plain upload."""
        X = np.ascontiguousarray(X)
        if X.ndim != 2:
            raise ValueError(f"Representation must be 2-D (got shape {X.shape}).")
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        # (NaN / infinity: screened by cm_rep_create on the device copy -> ValueError; numpy's isfinite cost more than the upload)
        self.h, self.shape, self.dtype = handle, X.shape, X.dtype
        r = C.c_void_p()
        handle.call("cm_rep_create_ex", _ptr(X), X.shape[0], X.shape[1], CM_F64 if X.dtype == np.float64 else CM_F32,
                    CM_REP_ALLOW_NAN if allow_nan else 0, C.byref(r))
        self._r = r

    def close(self):
        if getattr(self, "_r", None) and getattr(self.h, "_h", None):
            self.h.lib.cm_rep_destroy(self.h._h, self._r)
        self._r = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class DeviceGraph:
    """One k-NN graph kept in HBM (``cm_graph_*``), and after :meth:`mapping` its row-normalised mapping matrix: what the
    drop-in classes hold between ``compute_neighbors`` -> ``compute_mapping_matrix`` -> ``map_obs`` / ``map_obsm`` /
    ``map_layers``.  Host arrays are produced only by the ``fetch_*`` methods."""

    def __init__(self, handle: Handle, ref: DeviceRep, qry: DeviceRep, k: int, metric: str = "euclidean", form: str = "auto"):
        if metric not in METRICS:
            raise ValueError(f"Metric '{metric}' is not supported by the hip backend (supported: {SUPPORTED_METRICS_TEXT}).")
        if ref.shape[1] != qry.shape[1]:
            raise ValueError(f"Incompatible representation shapes {ref.shape} and {qry.shape}.")
        self.h = handle
        g = C.c_void_p()
        handle.call("cm_graph_knn", ref._r, qry._r, int(k), METRICS[metric], FORMS[form], C.byref(g))
        self._g = g
        self.nq, self.n_ref, self.k = qry.shape[0], ref.shape[0], int(k)
        self.dist_dtype = np.float32 if (METRICS[metric] == 1 and ref.dtype == np.float32) else np.float64
        self._cast_f32 = metric == "nan_euclidean" and ref.dtype == np.float32   # (sklearn returns float32 there; resident values are float64)
        self.nnz = None
        self.param = None
        self.map_tag = None

    def close(self):
        if getattr(self, "_g", None) and getattr(self.h, "_h", None):
            self.h.lib.cm_graph_destroy(self.h._h, self._g)
        self._g = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_devices(self) -> int:
        """GPUs the graph is spread over (query slices; > 1 only on a multi-device handle)."""
        out = np.zeros(8, np.int64)
        self.h.call("cm_graph_info", self._g, _ptr(out))
        return int(out[7])

    def parts(self):
        """One record per part (multi-device handle: the query slice a device owns): which reference rows that device
        searched, how many bytes of either representation it holds, how the per-shard lists were exchanged."""
        out = []
        for p in range(self.n_devices):
            v = np.zeros(8, np.int64)
            self.h.call("cm_graph_part_info", self._g, int(p), _ptr(v))
            out.append(dict(device=int(v[0]), q_lo=int(v[1]), nq=int(v[2]), ref_lo=int(v[3]), ref_rows=int(v[4]),
                            ref_bytes=int(v[5]), qry_bytes=int(v[6]), exchange=("peer copies", "host-staged", "none (query slices)")[int(v[7]) % 10],
                            self_check=("not run", "passed", "FAILED")[int(v[7]) // 10]))
        return out

    def fetch_knn(self):
        dist = np.empty((self.nq, self.k), self.dist_dtype)
        idx = np.empty((self.nq, self.k), np.int64)
        self.h.call("cm_graph_knn_fetch", self._g, _ptr(dist), _ptr(idx))
        if getattr(self, "_cast_f32", False):
            dist = dist.astype(np.float32)
        return dist, idx

    def mapping(self, kernel: str = "gaussian", epsilon: float = 1e-8):
        if kernel not in KERNELS:
            raise ValueError(
                f"Unknown kernel: {kernel}. Supported kernels are: 'gaussian', 'scarches', 'random', "
                "'inverse_distance', 'equal'.")
        nnz, param = C.c_int64(0), C.c_double(0.0)
        self.nnz = None
        self.map_tag = None
        self.h.call("cm_graph_mapping", self._g, KERNELS[kernel], float(epsilon), C.byref(nnz), C.byref(param))
        self.nnz, self.param = nnz.value, param.value
        # which kernel the RESIDENT mapping matrix was made with: a graph is shared between NeighborsResults and CellMapper,
        # and either may re-run the mapping in place (the holder of an older matrix checks this tag before using the graph)
        self.map_tag = (kernel, float(epsilon))
        return self.nnz, self.param

    @property
    def has_mapping(self) -> bool:
        return self.nnz is not None

    def fetch_mapping(self):
        indptr = np.empty(self.nq + 1, np.int64)
        cols = np.empty(self.nnz, np.int32)
        vals = np.empty(self.nnz, np.float32)
        self.h.call("cm_graph_mapping_fetch", self._g, _ptr(indptr), _ptr(cols), _ptr(vals))
        return indptr, cols, vals

    def label_vote(self, codes, n_cat: int):
        codes = np.ascontiguousarray(codes, np.int32)
        out_code = np.empty(self.nq, np.int32)
        out_conf = np.empty(self.nq, np.float32)
        self.h.call("cm_graph_label_vote", self._g, _ptr(codes), len(codes), int(n_cat), _ptr(out_code), _ptr(out_conf))
        return out_code, out_conf

    def spmm_dense(self, B):
        B = np.asarray(B)
        squeeze = B.ndim == 1
        if squeeze:
            B = B.reshape(-1, 1)
        if B.dtype != np.float32:
            B = B.astype(np.float64, copy=False)  # scipy upcast(float32, other) -> float64 for ints / float64
        B = np.ascontiguousarray(B)
        out = np.empty((self.nq, B.shape[1]), B.dtype)
        self.h.call("cm_graph_spmm_dense", self._g, _ptr(B), B.shape[0], B.shape[1], CM_F64 if B.dtype == np.float64 else CM_F32,
                    _ptr(out))
        return out[:, 0] if squeeze else out

    def spmm_csr(self, B_indptr, B_cols, B_vals, n_out: int):
        bi = np.ascontiguousarray(B_indptr, np.int64)
        bc = np.ascontiguousarray(B_cols, np.int32)
        bv = np.ascontiguousarray(B_vals, np.float32)
        n_ref = len(bi) - 1
        out_indptr = np.empty(self.nq + 1, np.int64)
        self.h.call("cm_graph_spmm_csr_nnz", self._g, _ptr(bi), _ptr(bc), _ptr(bv), n_ref, int(n_out), _ptr(out_indptr))
        nnz = int(out_indptr[self.nq])
        out_cols = np.empty(nnz, np.int32)
        out_vals = np.empty(nnz, np.float32)
        self.h.call("cm_graph_spmm_csr", self._g, _ptr(bi), _ptr(bc), _ptr(bv), n_ref, int(n_out), _ptr(out_indptr), _ptr(out_cols),
                    _ptr(out_vals))
        return out_indptr, out_cols, out_vals


_default = None
_default_lock = threading.Lock()


def default_devices():
    """Which GPUs the process-wide handle uses: ``CELLMAP_DEVICES`` ("all" or a comma-separated list) if set; the rank's own
    GPU under a ``torch.distributed`` / MPI launch (``LOCAL_RANK`` or ``CELLMAP_DEVICE`` set: one process per GPU); otherwise
    EVERY visible GPU -- ``CellMapper(query, reference).map(...)`` uses the whole node from one call."""
    env = os.environ.get("CELLMAP_DEVICES", "").strip()
    if env and env.lower() != "all":
        return [int(x) for x in env.split(",") if x.strip() != ""]
    if not env and ("CELLMAP_DEVICE" in os.environ or "LOCAL_RANK" in os.environ):
        return [int(os.environ.get("CELLMAP_DEVICE", os.environ.get("LOCAL_RANK", "0")))]
    n = int(load_library().cm_device_count())
    return list(range(max(n, 1)))


def reset_default_handle() -> None:
    """Destroy the process-wide handle (the next :func:`default_handle` creates a fresh one).  Objects that still hold device
    state of the old handle (``DeviceRep`` / ``DeviceGraph``) must be gone before this is called."""
    global _default
    with _default_lock:
        if _default is not None:
            _default.close()
            _default = None


def default_handle() -> Handle:
    """Process-wide handle on :func:`default_devices`, created on first use."""
    global _default
    with _default_lock:
        if _default is None:
            devs = default_devices()
            _default = Handle(devs[0]) if len(devs) == 1 else Handle(devices=devs)
        return _default
