#!/usr/bin/env python
"""Accuracy gate of the larger Winograd tiles, decided on the CPU before any kernel work (round 5 adds
F(3x3,3x3)): the float32 error of Winograd F(m x m,3x3) for several point sets against a float64 direct convolution, relative
to the error of a float32 DIRECT convolution, on one 128 -> 128 layer of SiLU-of-normal activations.  U = G g G^T is formed
in float64 and rounded once (as conv_pack_weights_wino does); V = B^T d B, the channel sum and A^T M A run in float32.
The gate: rms <= 2x the direct kernel's.   python tools/wino_accuracy.py   (profiles/r05_wino_accuracy.txt)"""
import numpy as np


def mats(points):
    """Cook-Toom matrices A^T, G, B^T for F(m, 3) from `points` (finite points + infinity), m = len(points) + 1 - 3 + ... """
    from fractions import Fraction
    import itertools
    n = len(points) + 1                     # tile size (with the point at infinity)
    m = n - 2
    pts = [Fraction(p) for p in points]
    # Lagrange / Vandermonde construction (Lavin & Gray): A^T [m x n], G [n x 3], B^T [n x n]
    AT = np.zeros((m, n)); G = np.zeros((n, 3)); 
    for i in range(m):
        for j, p in enumerate(pts):
            AT[i, j] = float(p ** i)
        AT[i, n - 1] = 1.0 if i == m - 1 else 0.0
    for j, p in enumerate(pts):
        denom = Fraction(1)
        for k, q in enumerate(pts):
            if k != j:
                denom *= (p - q)
        for i in range(3):
            G[j, i] = float(p ** i / denom)
    G[n - 1] = [0, 0, 1]
    # B^T from the polynomial identities: rows = coefficients of prod_{k != j}(x - p_k), last row = prod_k (x - p_k)
    BT = np.zeros((n, n))
    for j in range(n - 1):
        poly = np.poly1d([1.0])
        for k, q in enumerate(pts):
            if k != j:
                poly *= np.poly1d([1.0, -float(q)])
        c = poly.coeffs[::-1]
        BT[j, :len(c)] = c
    poly = np.poly1d([1.0])
    for q in pts:
        poly *= np.poly1d([1.0, -float(q)])
    c = poly.coeffs[::-1]
    BT[n - 1, :len(c)] = c
    return AT, G, BT


def winograd(x, w, AT, G, BT, dt):
    """x [C, H, W] (H, W multiples of m, zero padding 1), w [K, C, 3, 3] -> [K, H, W]; transforms and sums in `dt`."""
    m, n = AT.shape
    C, H, W = x.shape
    K = w.shape[0]
    U = np.einsum("ia,kcab,jb->ijkc", G, w.astype(np.float64), G).astype(dt)            # [n, n, K, C], rounded once
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1))).astype(dt)
    th, tw = H // m, W // m
    out = np.zeros((K, H, W), dtype=dt)
    BTd, ATd = BT.astype(dt), AT.astype(dt)
    for ty in range(th):
        d = np.stack([xp[:, ty * m: ty * m + n, tx * m: tx * m + n] for tx in range(tw)], 0)        # [tw, C, n, n]
        V = np.einsum("ia,tcab,jb->ijtc", BTd, d, BTd).astype(dt)
        # channel sum in dt, in order (pairs of channels as the MFMA would: close enough for an error estimate)
        M = np.zeros((n, n, tw, K), dtype=dt)
        for c in range(C):
            M += V[:, :, :, c, None] * U[:, :, None, :, c]
        Y = np.einsum("ia,abtk,jb->tkij", ATd, M, ATd).astype(dt)
        for tx in range(tw):
            out[:, ty * m:(ty + 1) * m, tx * m:(tx + 1) * m] = Y[tx]
    return out


def direct(x, w, dt):
    C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1))).astype(dt)
    out = np.zeros((w.shape[0], H, W), dtype=dt)
    wd = w.astype(dt)
    for c in range(C):
        for a in range(3):
            for b in range(3):
                out += wd[:, c, a, b][:, None, None] * xp[c, a:a + H, b:b + W][None]
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(7)
    C = K = 128
    H = W = 48
    x = rng.standard_normal((C, H, W)).astype(np.float32)
    x = x / (1.0 + np.exp(-x))                                        # SiLU of a normalised tensor, like the layers' inputs
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    ref = direct(x, w, np.float64)
    e_dir = direct(x, w, np.float32).astype(np.float64) - ref
    rows = [("direct f32", e_dir)]
    for name, pts in (("F(2x2,3x3) points 0, 1, -1   (shipped)", [0, 1, -1]),
                      ("F(3x3,3x3) points 0, 1, -1, 2", [0, 1, -1, 2]),
                      ("F(3x3,3x3) points 0, 1, -1, 1/2", [0, 1, -1, 0.5]),
                      ("F(3x3,3x3) points 0, 1, -1, -1/2", [0, 1, -1, -0.5]),
                      ("F(3x3,3x3) points 0, 1/2, -1/2, 1", [0, 0.5, -0.5, 1]),
                      ("F(4x4,3x3) points 0, 1, -1, 2, -2", [0, 1, -1, 2, -2]),
                      ("F(4x4,3x3) points 0, 1, -1, 1/2, -1/2", [0, 1, -1, 0.5, -0.5])):
        AT, G, BT = mats(pts)
        chk = winograd(x[:4, :12, :12].astype(np.float64), w[:4, :4].astype(np.float64), AT, G, BT, np.float64) - direct(x[:4, :12, :12], w[:4, :4], np.float64)
        assert np.abs(chk).max() < 1e-9, (name, np.abs(chk).max())     # the matrices are right
        e = winograd(x, w, AT, G, BT, np.float32).astype(np.float64) - ref
        rows.append((name, e))
    base, bmax = np.sqrt((rows[0][1] ** 2).mean()), np.abs(rows[0][1]).max()
    for name, e in rows:
        print("%-42s rms %.3e  max %.3e   rms / direct %.2f   max / direct %.2f" % (
            name, np.sqrt((e ** 2).mean()), np.abs(e).max(), np.sqrt((e ** 2).mean()) / base, np.abs(e).max() / bmax))
