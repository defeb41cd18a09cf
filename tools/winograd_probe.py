#!/usr/bin/env python3
"""Numerics probe for a possible next step: 1-D Winograd / Toom-Cook convolution F(m, r) for the MRF
tap counts.  The split kernels are power limited, so the one lever left is fewer multiplies per output: F(m, r) spends
m + r - 1 multiplies per m outputs instead of m * r.  This script measures what that costs in accuracy when the
element-wise products run with fp32 inputs / fp32 accumulation over the channels (as the MFMA path would), against an
fp64 direct convolution, on random data at the layers' scales.   python tools/winograd_probe.py"""
import itertools

import numpy as np


def toom_cook_matrices(m, r, points):
    """AT (m x n), G (n x r), BT (n x n) with n = m + r - 1 for interpolation points `points` (n - 1 finite points + infinity),
    built in exact rational arithmetic via fractions, returned as float64."""
    from fractions import Fraction as Fr
    n = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == n - 1

    def vander(rows, cols, with_inf):
        M = [[pts[i] ** j for j in range(cols)] for i in range(rows - (1 if with_inf else 0))]
        if with_inf:
            M.append([Fr(0)] * (cols - 1) + [Fr(1)])
        return M
    # y = AT [ (G g) * (BT d) ]:  A = Vandermonde(n x m) incl. infinity row, G = Vandermonde(n x r) scaled, B from the inverse
    A = vander(n, m, True)
    Gm = vander(n, r, True)
    # scale rows of G by 1 / prod_{j != i} (p_i - p_j)  (Lagrange denominators); infinity row unscaled
    for i in range(n - 1):
        den = Fr(1)
        for j in range(n - 1):
            if j != i:
                den *= (pts[i] - pts[j])
        Gm[i] = [v / den for v in Gm[i]]
    # BT: rows i < n-1: coefficients of prod_{j != i} (x - p_j) ... ; last row: coefficients of prod_j (x - p_j)
    def poly_mul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    BT = []
    for i in range(n - 1):
        poly = [Fr(1)]
        for j in range(n - 1):
            if j != i:
                poly = poly_mul(poly, [-pts[j], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for j in range(n - 1):
        poly = poly_mul(poly, [-pts[j], Fr(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(v) for v in row] for row in M], dtype=np.float64)  # noqa: E731
    return f(A).T, f(Gm), f(BT)


def check_exact(m, r, points):
    AT, G, BT = toom_cook_matrices(m, r, points)
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal(m + r - 1), rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([np.dot(d[i:i + r], g) for i in range(m)])
    return float(np.abs(y - ref).max())


def probe(m, r, points, C=128, T=4096, seed=1):
    """Conv over C channels: y[t] = sum_c sum_j w[c, j] x[c, t + j].  Winograd with fp32 transforms / products / channel
    accumulation vs direct fp32 vs fp64."""
    AT, G, BT = toom_cook_matrices(m, r, points)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((C, T + r - 1)).astype(np.float32)
    w = (rng.standard_normal((C, r)) / np.sqrt(C * r)).astype(np.float32)
    ref = np.zeros(T)
    for j in range(r):
        ref += (w[:, j:j + 1].astype(np.float64) * x[:, j:j + T].astype(np.float64)).sum(0)
    direct = np.zeros(T, dtype=np.float32)
    for j in range(r):
        direct += (w[:, j:j + 1] * x[:, j:j + T]).sum(0, dtype=np.float32)
    n = m + r - 1
    U = (w.astype(np.float64) @ G.T).astype(np.float32)             # (C, n) transformed weights (host, once)
    nt = T // m
    idx = (np.arange(nt)[:, None] * m + np.arange(n)[None, :])        # (nt, n)
    tiles = x[:, idx]                                                 # (C, nt, n)
    V = (tiles @ BT.T.astype(np.float32)).astype(np.float32)          # input transform in fp32
    M = (U[:, None, :] * V).sum(0, dtype=np.float32)                  # (nt, n): products + channel accumulation in fp32
    y = (M @ AT.T.astype(np.float32)).astype(np.float32).reshape(-1)  # output transform in fp32
    scale = np.abs(ref).max()
    return float(np.abs(y[: nt * m] - ref[: nt * m]).max() / scale), float(np.abs(direct - ref).max() / scale)


if __name__ == "__main__":
    cases = [(2, 3, [0, 1, -1]), (4, 3, [0, 1, -1, 2, -2]), (2, 7, [0, 1, -1, 2, -2, 0.5, -0.5]),
             (2, 11, [0, 1, -1, 2, -2, 0.5, -0.5, 3, -3, 1 / 3, -1 / 3]), (4, 7, [0, 1, -1, 2, -2, 0.5, -0.5, 3, -3])]
    print("F(m,r)  multiplies/output (direct r)   exactness(fp64)   winograd fp32 err / max|y|   direct fp32 err")
    for m, r, pts in cases:
        e0 = check_exact(m, r, pts)
        ew, ed = probe(m, r, pts)
        print(f"F({m},{r:2d})  {(m + r - 1) / m:5.2f} vs {r:2d}  ({r * m / (m + r - 1):.2f}x fewer)   {e0:.1e}   {ew:.2e}   {ed:.2e}")
