#!/usr/bin/env python
"""Numerical study that chose the interpolation points of the Winograd F(4x4,3x3) kernel (csrc/conv_wino4.hip): fp32 emulation in NumPy
(weights transformed in fp64 and rounded once, input / output transforms and the channel sums in fp32, sums in chunks of 4 like the
MFMA) against an fp64 direct convolution, post-ReLU inputs, fan-in scaled weights.  CPU only:  python tools/f4_accuracy_study.py"""
import numpy as np, sys
from fractions import Fraction
def cook_toom(points, m, r):
    # returns AT (m x a), G (a x r), BT (a x a) for F(m,r) with a=m+r-1 points incl. infinity, via Vandermonde construction (wincnn-like)
    import sympy
    from sympy import Rational, Matrix, symbols, Poly
    a = m + r - 1
    pts = [Rational(p) for p in points]  # a-1 finite points
    n = a
    x = symbols('x')
    # f = prod (x - p_i)
    def At(n_, m_):
        return Matrix(m_, n_, lambda i, j: (pts[j] ** i if j < n_ - 1 else (1 if i == m_ - 1 else 0)))
    AT = At(a, m)
    # G: a x r
    # F_i = prod_{j != i} (p_i - p_j)
    Fd = [sympy.prod([pts[i] - pts[j] for j in range(a - 1) if j != i]) for i in range(a - 1)]
    G = Matrix(a, r, lambda i, j: (pts[i] ** j / Fd[i] if i < a - 1 else (1 if j == r - 1 else 0)))
    # BT: from Lagrange polys
    f = sympy.prod([x - p for p in pts])
    BT = sympy.zeros(a, a)
    for i in range(a - 1):
        li = sympy.expand(f / (x - pts[i]))
        # scaled so that evaluation... use T matrix approach
        coeffs = Poly(li, x).all_coeffs()[::-1]
        coeffs += [0] * (a - len(coeffs))
        for j in range(a): BT[i, j] = coeffs[j]
    coeffs = Poly(sympy.expand(f), x).all_coeffs()[::-1]
    for j in range(a): BT[a - 1, j] = coeffs[j]
    return np.array(AT.tolist(), dtype=np.float64), np.array(G.tolist(), dtype=np.float64), np.array(BT.tolist(), dtype=np.float64)

def verify(AT, G, BT, m, r):
    rs = np.random.RandomState(0)
    d = rs.randn(m + r - 1); g = rs.randn(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()

def conv_direct64(x, w):
    # x [C,H,W] padded, w [O,C,3,3]; out [O,H-2,W-2]
    C, H, W = x.shape; O = w.shape[0]
    out = np.zeros((O, H - 2, W - 2))
    for dy in range(3):
        for dx in range(3):
            out += np.einsum('oc,chw->ohw', w[:, :, dy, dx], x[:, dy:dy + H - 2, dx:dx + W - 2])
    return out

def wino(x, w, AT, G, BT, m):
    a = BT.shape[0]
    C, H, W = x.shape; O = w.shape[0]
    Ho, Wo = H - 2, W - 2
    ty, tx = Ho // m, Wo // m
    U = np.einsum('ij,ocjk,lk->ocil', G, w.astype(np.float64), G).astype(np.float32)   # fp64 transform, rounded once
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    out = np.zeros((O, Ho, Wo), np.float32)
    x32 = x.astype(np.float32)
    # tiles
    d = np.zeros((C, ty, tx, a, a), np.float32)
    for i in range(ty):
        for j in range(tx):
            d[:, i, j] = x32[:, i * m:i * m + a, j * m:j * m + a]
    # input transform in fp32: V = BT d B  (do as two fp32 matmuls)
    V = np.einsum('ij,ctujk->ctuik', BT32, d).astype(np.float32)
    V = np.einsum('ctuik,lk->ctuil', V, BT32).astype(np.float32)
    # per position GEMM over C in fp32 with sequential chunks of 4 (MFMA k=4 accumulate chain)
    M = np.zeros((O, ty, tx, a, a), np.float32)
    for c0 in range(0, C, 4):
        M += np.einsum('ocil,ctuil->otuil', U[:, c0:c0 + 4], V[c0:c0 + 4]).astype(np.float32)
    Y = np.einsum('ij,otujk->otuik', AT32, M).astype(np.float32)
    Y = np.einsum('otuik,lk->otuil', Y, AT32).astype(np.float32)
    for i in range(ty):
        for j in range(tx):
            out[:, i * m:(i + 1) * m, j * m:(j + 1) * m] = Y[:, i, j]
    return out

if __name__ == "__main__":
    sets = {"F2 (0,1,-1)": ([0, 1, -1], 2), "F4 lavin (0,1,-1,2,-2)": ([0, 1, -1, 2, -2], 4),
            "F4 (0,1,-1,1/2,-1/2)": ([0, 1, -1, Fraction(1, 2), Fraction(-1, 2)], 4),
            "F4 (0,1,-1,1/2,-2)": ([0, 1, -1, Fraction(1, 2), -2], 4),
            "F4 (0,1,-1,2,-1/2)": ([0, 1, -1, 2, Fraction(-1, 2)], 4)}
    rs = np.random.RandomState(1)
    for (C, O, HW) in [(64, 64, 24), (256, 256, 24), (512, 128, 24)]:
        x = np.maximum(rs.randn(C, HW + 2, HW + 2), 0)   # post-ReLU activations
        x[:, 0, :] = x[:, -1, :] = 0; x[:, :, 0] = x[:, :, -1] = 0
        b = (6.0 / (C * 9)) ** 0.5
        w = rs.uniform(-b, b, (O, C, 3, 3))
        ref = conv_direct64(x, w)
        # direct fp32 for comparison
        d32 = np.zeros_like(ref, dtype=np.float32)
        for dy in range(3):
            for dx in range(3):
                for c0 in range(0, C, 4):
                    d32 += np.einsum('oc,chw->ohw', w[:, c0:c0+4, dy, dx].astype(np.float32), x[c0:c0+4, dy:dy + HW, dx:dx + HW].astype(np.float32)).astype(np.float32)
        print("C=%d O=%d: direct fp32 err/max %.2e" % (C, O, np.abs(d32 - ref).max() / np.abs(ref).max()))
        for name, (pts, m) in sets.items():
            AT, G, BT = cook_toom(pts, m, 3)
            assert verify(AT, G, BT, m, 3) < 1e-9, name
            y = wino(x, w, AT, G, BT, m)
            print("   %-28s err/max %.2e   rms/max %.2e" % (name, np.abs(y - ref).max() / np.abs(ref).max(), np.sqrt(((y - ref) ** 2).mean()) / np.abs(ref).max()))
