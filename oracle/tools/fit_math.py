#!/usr/bin/env python3
"""Fits the polynomial coefficients used by vkr_math.h / oracle_math.h and
reports their accuracy against double-precision libm (in float ulps).

The same coefficients are typed into both headers; this script is the
provenance.  Evaluation below mimics the float arithmetic (explicit fma via
float64 emulation is not exact, so the final ulp numbers are re-measured in C by
tests/test_oracle_math.py)."""
import numpy as np
from numpy.polynomial import chebyshev as C


def fit_minimax_rel(f, lo, hi, deg, n=4000, iters=60):
    """Weighted least squares on Chebyshev nodes, iteratively re-weighted towards
    minimax (Lawson).  Returns monomial coefficients (lowest first)."""
    k = np.arange(n)
    x = 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * (k + 0.5) / n)
    y = f(x)
    w = np.ones_like(x)
    V = np.vander(x, deg + 1, increasing=True)
    for _ in range(iters):
        sw = np.sqrt(w)
        c, *_ = np.linalg.lstsq(V * sw[:, None], y * sw, rcond=None)
        err = np.abs(V @ c - y)
        w = w * (err + 1e-300)
        w /= w.sum()
    return c


def f32(c):
    return [float(np.float32(v)) for v in c]


def show(name, c):
    print(name + " = {" + ", ".join("%.9ef" % v for v in f32(c)) + "}")


# atan(z) = z + z*s*P(s), s = z^2, z in [0,1]
def atan_target(s):
    z = np.sqrt(np.maximum(s, 1e-300))
    taylor = -1.0 / 3.0 + s / 5.0 - s * s / 7.0 + s ** 3 / 9.0
    return np.where(s > 1e-3, (np.arctan(z) / z - 1.0) / np.maximum(s, 1e-300), taylor)

ca = fit_minimax_rel(atan_target, 0.0, 1.0, 8)
show("ATAN_P", ca)

# asin(z) = z + z*s*R(s), s = z^2 in [0, 0.25]
def asin_target(s):
    z = np.sqrt(np.maximum(s, 1e-300))
    taylor = 1.0 / 6.0 + 3.0 * s / 40.0 + 15.0 * s * s / 336.0
    return np.where(s > 1e-3, (np.arcsin(z) / z - 1.0) / np.maximum(s, 1e-300), taylor)

cs = fit_minimax_rel(asin_target, 0.0, 0.25, 5)
show("ASIN_R", cs)

# sin(r) = r + r*s*S(s), cos(r) = 1 - s/2 + s*s*Cc(s), |r| <= pi/4
def sin_target(s):
    r = np.sqrt(np.maximum(s, 1e-300))
    taylor = -1.0 / 6.0 + s / 120.0 - s * s / 5040.0
    return np.where(s > 1e-3, (np.sin(r) / r - 1.0) / np.maximum(s, 1e-300), taylor)

def cos_target(s):
    r = np.sqrt(np.maximum(s, 1e-300))
    taylor = 1.0 / 24.0 - s / 720.0 + s * s / 40320.0 - s ** 3 / 3628800.0
    return np.where(s > 1e-2, (np.cos(r) - 1.0 + 0.5 * s) / np.maximum(s * s, 1e-300), taylor)

smax = (np.pi / 4) ** 2 * 1.01
show("SIN_S", fit_minimax_rel(sin_target, 0.0, smax, 3))
show("COS_C", fit_minimax_rel(cos_target, 0.0, smax, 3))
