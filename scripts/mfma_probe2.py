#!/usr/bin/env python3
"""Designed inputs for v_mfma_f32_32x32x16_f16: how many bits below the largest addend survive the internal alignment?"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from cellmapper_amd import _native
h = _native.default_handle()

def run(a, b, c):
    n = a.shape[0]
    d = np.empty_like(c)
    h.call("cm_debug_mfma_f16", a.view(np.uint16).ctypes.data_as(C.c_void_p), b.view(np.uint16).ctypes.data_as(C.c_void_p),
           c.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), n, 1)
    return d

S = list(range(18, 30))
n = len(S)
# case A: C = 1, sixteen products 2^-s each (exact sum 2^(4-s))
a = np.zeros((n, 1, 32, 16), np.float16); b = np.zeros((n, 1, 16, 32), np.float16); c = np.ones((n, 32, 32), np.float32)
for t, s in enumerate(S):
    a[t, 0, :, :] = np.float16(2.0 ** -(s // 2)); b[t, 0, :, :] = np.float16(2.0 ** -(s - s // 2))
d = run(a, b, c)
print("A: C=1 + 16 x 2^-s   -> (D-1)/2^-23 :", {s: float((d[t, 0, 0] - 1.0) / 2.0 ** -23) for t, s in enumerate(S)})
# case B: C = 0, one product 1.0 and fifteen products 2^-s
a[:] = 0; b[:] = 0; c[:] = 0
for t, s in enumerate(S):
    a[t, 0, :, :] = np.float16(2.0 ** -(s // 2)); b[t, 0, :, :] = np.float16(2.0 ** -(s - s // 2))
    a[t, 0, :, 0] = 1.0; b[t, 0, 0, :] = 1.0
d = run(a, b, c)
print("B: 1 + 15 x 2^-s (C=0) -> (D-1)/2^-23 :", {s: float((d[t, 0, 0] - 1.0) / 2.0 ** -23) for t, s in enumerate(S)})
# case C: C = 1, one product (1 + 2^-10) * 2^-s' : does a product keep its own low bits?
a[:] = 0; b[:] = 0; c[:] = 1
for t, s in enumerate(S):
    a[t, 0, :, 0] = np.float16(1.0 + 2.0 ** -10); b[t, 0, 0, :] = np.float16(2.0 ** -(s - 10))
d = run(a, b, c)
print("C: C=1 + (1+2^-10) 2^-(s-10) -> (D-1)/2^-23 :", {s: float((d[t, 0, 0] - 1.0) / 2.0 ** -23) for t, s in enumerate(S)})
# case D: cancellation: C = -1, products: 1.0 and 2^-s  (exact result 2^-s)
a[:] = 0; b[:] = 0; c[:] = -1
for t, s in enumerate(S):
    a[t, 0, :, 0] = 1.0; b[t, 0, 0, :] = 1.0
    a[t, 0, :, 1] = np.float16(2.0 ** -(s // 2)); b[t, 0, 1, :] = np.float16(2.0 ** -(s - s // 2))
d = run(a, b, c)
print("D: -1 + 1 + 2^-s -> D * 2^s :", {s: float(d[t, 0, 0] * 2.0 ** s) for t, s in enumerate(S)})
