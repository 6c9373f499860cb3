import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/cellmapper_amd") else os.getcwd())
import torch  # noqa
from cellmapper_amd import _native
h = _native.default_handle()
rng = np.random.default_rng(1)
def run(a, b, c, chain):
    n = c.shape[0]; d = np.empty_like(c)
    h.call("cm_debug_mfma_f16", a.view(np.uint16).ctypes.data_as(C.c_void_p), b.view(np.uint16).ctypes.data_as(C.c_void_p),
           c.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), n, chain)
    return d
for chain in (1, 4, 12, 24):
    worst = 0.0
    for spread in (0, 3, 8):
        n = 300
        ea = rng.integers(-spread, spread + 1, (n, chain, 32, 16)); eb = rng.integers(-spread, spread + 1, (n, chain, 16, 32))
        a = (rng.standard_normal((n, chain, 32, 16)) * np.exp2(ea)).astype(np.float16)
        b = (rng.standard_normal((n, chain, 16, 32)) * np.exp2(eb)).astype(np.float16)
        c = (rng.standard_normal((n, 32, 32)) * np.exp2(rng.integers(-4, 8, (n, 32, 32)))).astype(np.float32)
        d = run(a, b, c, chain)
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        exact = c.astype(np.float64) + np.einsum("nsik,nskj->nij", a64, b64)
        S = np.abs(c.astype(np.float64)) + np.einsum("nsik,nskj->nij", np.abs(a64), np.abs(b64))
        r = np.abs(d.astype(np.float64) - exact) / S
        worst = max(worst, r.max())
        print(f"chain {chain:2d} spread {spread}: max err/S = {r.max():.3e} = {r.max()/2**-24:.2f} x 2^-24 ; bound (16+2n) = {16+2*chain}")
