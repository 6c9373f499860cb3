#!/usr/bin/env python3
"""Which arithmetic does v_mfma_f32_32x32x16_f16 implement?  Runs random problems through cm_debug_mfma_f16 and compares
the float32 results bit for bit with exact-rational models of the accumulation (groupings of the 16 products, RNE / RTZ)."""
import os, sys
from fractions import Fraction
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rnd32(x: Fraction, mode="rne") -> np.float32:
    """exact rounding of a rational to float32 (normal/subnormal range; no overflow handling needed here)"""
    if x == 0:
        return np.float32(0.0)
    s = -1 if x < 0 else 1
    ax = abs(x)
    e = ax.numerator.bit_length() - ax.denominator.bit_length()  # 2^(e-1) <= ax < 2^(e+1)
    if Fraction(2) ** e > ax:
        e -= 1
    # now 2^e <= ax < 2^(e+1); float32 ulp = 2^(e-23), but not below 2^-149
    q = max(e - 23, -149)
    m = ax / (Fraction(2) ** q)
    fl = m.numerator // m.denominator
    rem = m - fl
    if mode == "rne":
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
            fl += 1
    return np.float32(s * float(fl) * (2.0 ** q))


def model(c, prods, group, mode):
    t = Fraction(float(c))
    for g0 in range(0, 16, group):
        t = Fraction(float(rnd32(t + sum(prods[g0:g0 + group], Fraction(0)), mode)))
    return np.float32(float(t))


def main():
    import torch  # noqa: F401  (shares the HIP runtime with the library)
    from cellmapper_amd import _native
    h = _native.default_handle()
    rng = np.random.default_rng(0)
    n = int(os.environ.get("PROBE_N", "6"))
    # wide dynamic range so that the rounding points differ between models
    a = (rng.standard_normal((n, 1, 32, 16)) * np.exp2(rng.integers(-6, 7, (n, 1, 32, 16)))).astype(np.float16)
    b = (rng.standard_normal((n, 1, 16, 32)) * np.exp2(rng.integers(-6, 7, (n, 1, 16, 32)))).astype(np.float16)
    c = (rng.standard_normal((n, 32, 32)) * np.exp2(rng.integers(-4, 12, (n, 32, 32)))).astype(np.float32)
    d = np.empty_like(c)
    import ctypes as C
    h.call("cm_debug_mfma_f16", a.view(np.uint16).ctypes.data_as(C.c_void_p), b.view(np.uint16).ctypes.data_as(C.c_void_p),
           c.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), n, 1)
    models = [(g, m) for g in (16, 8, 4, 2, 1) for m in ("rne", "rtz")]
    hits = {k: 0 for k in models}
    total = 0
    worst = 0.0
    for p in range(n):
        for i in range(0, 32, 3):
            for j in range(0, 32, 5):
                prods = [Fraction(float(a[p, 0, i, k])) * Fraction(float(b[p, 0, k, j])) for k in range(16)]
                exact = Fraction(float(c[p, i, j])) + sum(prods, Fraction(0))
                mag = abs(Fraction(float(c[p, i, j]))) + sum((abs(x) for x in prods), Fraction(0))
                worst = max(worst, float(abs(Fraction(float(d[p, i, j])) - exact) / mag))
                total += 1
                for k in models:
                    if model(c[p, i, j], prods, k[0], k[1]).view(np.uint32) == d[p, i, j].view(np.uint32):
                        hits[k] += 1
    print("problems", n, "entries checked", total)
    for k in models:
        print(f"group={k[0]:2d} {k[1]}: {hits[k]}/{total} bit-exact")
    print("max |D - exact| / (|C| + sum|a b|) = %.3e  (2^-24 = 5.96e-8)" % worst)


main()
