"""Device tier plumbing: call the C ABI's ``cm_*_dev`` entry points on torch-owned device memory.

torch is used for what the task allows it for -- device buffers, streams and ``torch.distributed`` -- while all
compute goes through ``libcellmap_hip.so``.  Tensors must live on the handle's GPU and be contiguous.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native
from ._native import CM_F32, CM_F64, FORMS, KERNELS, METRICS, _ptr


def _dp(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class DevicePipeline:
    """One GPU's worth of the hot path on device-resident tensors."""

    def __init__(self, device_index: int = 0, use_torch_stream: bool = True):
        import torch

        self.torch = torch
        self.device = torch.device("cuda", device_index)
        self.h = _native.Handle(device_index)
        # a dedicated torch stream carries BOTH the library's kernels and torch's own ops / RCCL collectives, so
        # ordering between them is plain stream order.  Use `with torch.cuda.stream(pipe.stream): ...`.
        self.stream = torch.cuda.Stream(self.device) if use_torch_stream else None
        if self.stream is not None:
            self.h.set_stream(self.stream.cuda_stream)

    def synchronize(self):
        """Wait for the stream and surface what the asynchronous device tier could not return as a status: a query whose
        exact-fallback list overflowed has a truncated neighbour list (cm_knn_stats out[4])."""
        self.torch.cuda.synchronize(self.device)
        if self.h.knn_stats()["fallback_overflow"]:
            raise NotImplementedError("k-NN exact fallback overflow: more than 4096 reference rows tie within a query's "
                                      "k-th distance")

    def _dtype_id(self, t):
        if t.dtype == self.torch.float32:
            return CM_F32
        if t.dtype == self.torch.float64:
            return CM_F64
        raise ValueError(f"unsupported dtype {t.dtype}")

    def knn(self, ref, qry, k, metric="euclidean", form="auto", idx_offset=0, squared=False, force_fallback=False,
            f64_out=False, out=None):
        """``out=(dist, idx)``: caller-provided contiguous [nq, k] tensors (e.g. planes of a packed exchange buffer)."""
        t = self.torch
        nq = qry.shape[0]
        out_dt = t.float64 if (METRICS[metric] != 1 or f64_out) else ref.dtype   # (sklearn: float32 cosine stays float32)
        if out is None:
            dist = t.empty((nq, k), dtype=out_dt, device=self.device)
            idx = t.empty((nq, k), dtype=t.int64, device=self.device)
        else:
            dist, idx = out
            if dist.dtype != out_dt or idx.dtype != t.int64 or tuple(dist.shape) != (nq, k) or tuple(idx.shape) != (nq, k) \
                    or not (dist.is_contiguous() and idx.is_contiguous()):
                raise ValueError("knn(out=...): need contiguous [nq, k] tensors (distance dtype %s, indices int64)" % out_dt)
        flags = (1 if squared else 0) | (2 if force_fallback else 0) | (4 if f64_out else 0)
        self.h.call("cm_knn_dev", _dp(ref), ref.shape[0], _dp(qry), nq, ref.shape[1], self._dtype_id(ref), int(k),
                    METRICS[metric], FORMS[form], int(idx_offset), flags, _dp(dist), _dp(idx))
        return dist, idx

    def merge(self, dist_parts, idx_parts, k, sqrt_out=True):
        """dist_parts / idx_parts: [nparts, nq, k] (squared distances if sqrt_out) -> merged [nq, k]."""
        t = self.torch
        nparts, nq, _ = dist_parts.shape
        od = t.empty((nq, k), dtype=t.float64, device=self.device)
        oi = t.empty((nq, k), dtype=t.int64, device=self.device)
        self.h.call("cm_knn_merge_dev", _dp(dist_parts), _dp(idx_parts), int(nparts), int(nq), int(k),
                    1 if sqrt_out else 0, _dp(od), _dp(oi))
        return od, oi

    def merge_strided(self, buf, nparts, part_stride, dist_off, idx_off, nq, k, sqrt_out=True):
        """Merge ``nparts`` lists per query out of ONE packed buffer of 8-byte words (the received exchange buffer):
        part p's distance rows start at word ``dist_off + p*part_stride``, its index rows at ``idx_off + p*part_stride``."""
        t = self.torch
        if buf.element_size() != 8 or not buf.is_contiguous():
            raise ValueError("merge_strided: need a contiguous buffer of 8-byte words")
        need = (nparts - 1) * part_stride + max(dist_off, idx_off) + nq * k
        if need > buf.numel():
            raise ValueError(f"merge_strided: buffer of {buf.numel()} words is too small (need {need})")
        od = t.empty((nq, k), dtype=t.float64, device=self.device)
        oi = t.empty((nq, k), dtype=t.int64, device=self.device)
        base = buf.data_ptr()
        self.h.call("cm_knn_merge_strided_dev", C.c_void_p(base + 8 * dist_off), C.c_void_p(base + 8 * idx_off),
                    int(nparts), int(part_stride), int(nq), int(k), 1 if sqrt_out else 0, _dp(od), _dp(oi))
        return od, oi

    def moments(self, dist, idx, center=None):
        m = self.torch.zeros(3, dtype=self.torch.float64, device=self.device)
        self.h.call("cm_dist_moments_dev", _dp(dist), _dp(idx), dist.numel(), _dp(center), _dp(m))
        return m

    def kernel_param(self, moments3, kernel):
        p = self.torch.zeros(1, dtype=self.torch.float64, device=self.device)
        self.h.call("cm_kernel_param_dev", _dp(moments3), KERNELS[kernel], _dp(p))
        return p

    def weights(self, dist, idx, n_ref, kernel, param, epsilon=1e-8):
        t = self.torch
        nq, k = dist.shape
        cols = t.empty((nq, k), dtype=t.int32, device=self.device)
        vals = t.empty((nq, k), dtype=t.float32, device=self.device)
        cnt = t.empty(nq, dtype=t.int32, device=self.device)
        self.h.call("cm_weights_dev", _dp(dist), _dp(idx), nq, k, int(n_ref), KERNELS[kernel], _dp(param),
                    float(epsilon), _dp(cols), _dp(vals), _dp(cnt), None)
        return cols, vals, cnt

    def to_csr(self, cols, vals, cnt):
        t = self.torch
        nq, k = cols.shape
        indptr = t.empty(nq + 1, dtype=t.int64, device=self.device)
        oc = t.empty(nq * k, dtype=t.int32, device=self.device)
        ov = t.empty(nq * k, dtype=t.float32, device=self.device)
        self.h.call("cm_ell_to_csr_dev", _dp(cols), _dp(vals), _dp(cnt), nq, k, _dp(indptr), _dp(oc), _dp(ov))
        return indptr, oc, ov

    def label_vote(self, indptr, cols, vals, codes, n_cat):
        t = self.torch
        nq = indptr.shape[0] - 1
        code = t.empty(nq, dtype=t.int32, device=self.device)
        conf = t.empty(nq, dtype=t.float32, device=self.device)
        self.h.call("cm_label_vote_dev", _dp(indptr), _dp(cols), _dp(vals), nq, _dp(codes), codes.shape[0], int(n_cat),
                    _dp(code), _dp(conf))
        return code, conf

    def spmm_dense(self, indptr, cols, vals, B, out=None):
        """``M @ B`` for a dense payload (A1 / A2).  Query tiles: pass ``indptr[q0:q1 + 1]`` (a view)."""
        t = self.torch
        nq = indptr.shape[0] - 1
        if out is None:
            out = t.empty((nq, B.shape[1]), dtype=B.dtype, device=self.device)
        elif tuple(out.shape) != (nq, B.shape[1]) or out.dtype != B.dtype or not out.is_contiguous():
            raise ValueError("spmm_dense(out=...): need a contiguous [nq, n_out] tensor of the payload's dtype")
        self.h.call("cm_spmm_dense_dev", _dp(indptr), _dp(cols), _dp(vals), nq, _dp(B), B.shape[0], B.shape[1],
                    self._dtype_id(B), _dp(out))
        return out

    def locality_order(self, indptr, cols, n_ref):
        """Permutation of the mapping-matrix rows (int32) in which rows sharing their smallest neighbour column are adjacent:
        a processing order for :meth:`spmm_csr` that keeps the gathered payload rows in L2."""
        t = self.torch
        nq = indptr.shape[0] - 1
        order = t.empty(nq, dtype=t.int32, device=self.device)
        self.h.call("cm_locality_order_dev", _dp(indptr), _dp(cols), nq, int(n_ref), _dp(order))
        return order

    SINGLE_PASS_MAX_COLS = 32768

    def payload_index(self, b_indptr, b_cols, n_out):
        """Index of a CSR payload for the single-pass :meth:`spmm_csr` (``cm_payload_index_dev``, built once per payload,
        synchronous): lets every wave of a workgroup own an eighth of the output columns, so the product needs no barrier
        per neighbour.  Raises ``ValueError`` if a payload row is not canonical CSR (strictly ascending columns)."""
        t = self.torch
        n_ref = b_indptr.shape[0] - 1
        n = int(self.h.lib.cm_payload_index_len(n_ref, int(n_out)))
        if n < 0:
            raise ValueError(f"payload_index: the single-pass product covers 1 .. {self.SINGLE_PASS_MAX_COLS} output columns")
        index = t.empty(max(n, 1), dtype=t.int32, device=self.device)
        self.h.call("cm_payload_index_dev", _dp(b_indptr), _dp(b_cols), n_ref, int(n_out), _dp(index))
        return index

    def spmm_csr(self, indptr, cols, vals, b_indptr, b_cols, b_vals, n_out, q0=0, q1=None, out=None, order=None,
                 single_pass=None, out_rows=None, payload_index=None):
        """Rows [q0, q1) of ``M @ B`` for a CSR payload (A3, device tier): ``(out_indptr, out_cols, out_vals)`` of the TILE
        (``out_indptr`` starts at 0).  Without ``out``: count pass, exclusive scan, exact allocation, fill pass -- the
        structure of scipy's csr_matmat.  ``out=(ptr [>= nt+1] int64, cols int32, vals float32)``: caller-provided tile
        buffers with room for the worst case (``nt * min(n_out, k * max payload row)`` entries) -- no host read at all, the
        tile's entry count is ``ptr[nt]`` on the device; with ``n_out <= 32 768`` the product is then made in ONE pass
        (``cm_spmm_csr_fused_dev``: offsets from a chained scan inside the kernel; rows that would end beyond the buffers
        are counted but not written).  ``single_pass=False`` forces the two-pass pair.  ``out_rows`` (int32 [>= nt], single
        pass only): rows are emitted in COMPLETION order, output row j = product row of tile row ``out_rows[j]`` -- no
        workgroup waits for its predecessors' counts (the fastest form; the arrangement differs from run to run, the
        content of a row does not).  ``payload_index`` (from :meth:`payload_index`, single pass only): the wave-sliced
        kernel -- no barrier per neighbour, same bits."""
        t = self.torch
        nq_all = indptr.shape[0] - 1
        q1 = nq_all if q1 is None else q1
        nt = q1 - q0
        n_ref = b_indptr.shape[0] - 1
        # a tile is a slice of the indptr (a view: entries still index the full cols / vals), or -- with a row order -- a
        # slice of the ORDER over the full indptr: output row r = product row of mapping-matrix row order[q0 + r]
        ip = indptr[q0:q1 + 1] if order is None else indptr
        od = None if order is None else order[q0:q1]
        optr = out[0] if out is not None else t.empty(nt + 1, dtype=t.int64, device=self.device)
        if out is not None:
            oc, ov = out[1], out[2]
            if optr.dtype != t.int64 or oc.dtype != t.int32 or ov.dtype != t.float32 or optr.shape[0] < nt + 1:
                raise ValueError("spmm_csr(out=...): need (int64 [>= nt+1], int32, float32) device tensors")
            if single_pass is None:
                single_pass = int(n_out) <= self.SINGLE_PASS_MAX_COLS and not os.environ.get("CELLMAP_SPGEMM_CT") \
                    and not os.environ.get("CELLMAP_A3_TWO_PASS")
            if out_rows is not None and (out_rows.dtype != t.int32 or out_rows.shape[0] < nt):
                raise ValueError("spmm_csr(out_rows=...): need an int32 device tensor with one entry per tile row")
            if single_pass:
                self.h.call("cm_spmm_csr_fused_dev", _dp(ip), _dp(cols), _dp(vals), nt, _dp(b_indptr), _dp(b_cols), _dp(b_vals),
                            n_ref, int(n_out), int(min(oc.shape[0], ov.shape[0])), _dp(optr), _dp(oc), _dp(ov), _dp(od),
                            _dp(out_rows), _dp(payload_index))
                return optr, oc, ov
        elif single_pass:
            raise ValueError("spmm_csr(single_pass=True) needs caller-provided tile buffers (out=...)")
        if out_rows is not None:
            raise ValueError("spmm_csr(out_rows=...): completion order exists for the single-pass form only")
        self.h.call("cm_spmm_csr_nnz_dev", _dp(ip), _dp(cols), _dp(vals), nt, _dp(b_indptr), _dp(b_cols), _dp(b_vals), n_ref,
                    int(n_out), _dp(optr), _dp(od))
        if out is not None:
            nnz = None
        else:
            nnz = int(optr[-1].item())   # (one host read per tile: the tile's output has to be allocated)
            oc = t.empty(max(nnz, 1), dtype=t.int32, device=self.device)
            ov = t.empty(max(nnz, 1), dtype=t.float32, device=self.device)
        self.h.call("cm_spmm_csr_dev", _dp(ip), _dp(cols), _dp(vals), nt, _dp(b_indptr), _dp(b_cols), _dp(b_vals), n_ref,
                    int(n_out), _dp(optr), _dp(oc), _dp(ov), _dp(od))
        return (optr, oc, ov) if nnz is None else (optr, oc[:nnz], ov[:nnz])

    def map_labels(self, ref, qry, k, codes, n_cat, metric="euclidean", kernel="gaussian", epsilon=1e-8,
                   out_code=None, out_conf=None, want_graph=False):
        """Fused k-NN -> kernel -> normalise -> label vote, every intermediate resident in HBM."""
        t = self.torch
        nq = qry.shape[0]
        code = out_code if out_code is not None else t.empty(nq, dtype=t.int32, device=self.device)
        conf = out_conf if out_conf is not None else t.empty(nq, dtype=t.float32, device=self.device)
        dist = idx = None
        if want_graph:
            dist = t.empty((nq, k), dtype=t.float64, device=self.device)
            idx = t.empty((nq, k), dtype=t.int64, device=self.device)
        self.h.call("cm_map_labels_dev", _dp(ref), ref.shape[0], _dp(qry), nq, ref.shape[1], self._dtype_id(ref), int(k),
                    METRICS[metric], KERNELS[kernel], float(epsilon), _dp(codes), int(n_cat), _dp(code), _dp(conf),
                    _dp(dist), _dp(idx))
        return (code, conf, dist, idx) if want_graph else (code, conf)
