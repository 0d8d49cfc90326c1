#!/usr/bin/env python
"""Accuracy gate of the larger Winograd tiles, decided on the CPU before any kernel work (round 5 adds
F(3x3,3x3)): the float32 error of Winograd F(m x m,3x3) for several point sets against a float64 direct convolution, relative
to the error of a float32 DIRECT convolution, on one 128 -> 128 layer of SiLU-of-normal activations.  U = G g G^T is formed
in float64 and rounded once (as conv_pack_weights_wino does); V = B^T d B, the channel sum and A^T M A run in float32.
The gate: rms <= 2x the direct kernel's.   python tools/wino_accuracy.py [--up2]   (profiles/r05_wino_accuracy.txt;
--up2: the Upsample layer's three forms, round 5)"""
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


def winograd_mixed(x, w, mr, mc, dt):
    """F(m_r x m_c, 3x3) with different tiles along rows and columns (round 5: F(2,3) x F(3,3), 20 products per 6 outputs)."""
    ATr, Gr, BTr = mr
    ATc, Gc, BTc = mc
    m_r, n_r = ATr.shape
    m_c, n_c = ATc.shape
    C, H, W = x.shape
    K = w.shape[0]
    U = np.einsum("ia,kcab,jb->ijkc", Gr, w.astype(np.float64), Gc).astype(dt)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1))).astype(dt)
    th, tw = H // m_r, W // m_c
    out = np.zeros((K, H, W), dtype=dt)
    for ty in range(th):
        d = np.stack([xp[:, ty * m_r: ty * m_r + n_r, tx * m_c: tx * m_c + n_c] for tx in range(tw)], 0)
        V = np.einsum("ia,tcab,jb->ijtc", BTr.astype(dt), d, BTc.astype(dt)).astype(dt)
        M = np.zeros((n_r, n_c, tw, K), dtype=dt)
        for c in range(C):
            M += V[:, :, :, c, None] * U[:, :, None, :, c]
        Y = np.einsum("ia,abtk,jb->tkij", ATr.astype(dt), M, ATc.astype(dt)).astype(dt)
        for tx in range(tw):
            out[:, ty * m_r:(ty + 1) * m_r, tx * m_c:(tx + 1) * m_c] = Y[tx]
    return out


def upsample_forms():
    """The Upsample layer (nearest 2x + 3x3) three ways in float32 against float64: the 3x3 form on the up-sampled image, the four
    2x2-tap parity convolutions (conv_ws.hip), and those in the Winograd F(2x2,2x2) domain (conv_wup2.hip).   --up2"""
    rng = np.random.default_rng(7)
    C = K = 128; H = W = 24
    x = rng.standard_normal((C, H, W)).astype(np.float32)      # an Upsample's input is a ResidualBlock output (no activation)
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)

    def direct3x3_up(x, w, dt):
        xu = np.repeat(np.repeat(x, 2, 1), 2, 2)
        Cc, Hh, Ww = xu.shape
        xp = np.pad(xu, ((0, 0), (1, 1), (1, 1))).astype(dt)
        out = np.zeros((w.shape[0], Hh, Ww), dtype=dt); wd = w.astype(dt)
        for c in range(Cc):
            for a in range(3):
                for b in range(3):
                    out += wd[:, c, a, b][:, None, None] * xp[c, a:a + Hh, b:b + Ww][None]
        return out

    def par_w(w):        # [a][b][K][C][2][2] in double
        w = w.astype(np.float64)
        rows = {0: [[0], [1, 2]], 1: [[0, 1], [2]]}
        out = np.zeros((2, 2) + w.shape[:2] + (2, 2))
        for a in range(2):
            for b in range(2):
                for i in range(2):
                    for j in range(2):
                        out[a, b, :, :, i, j] = sum(w[:, :, ky, kx] for ky in rows[a][i] for kx in rows[b][j])
        return out

    def parity_direct(x, w, dt):
        pw = par_w(w).astype(dt)
        Cc, Hh, Ww = x.shape
        xp = np.pad(x, ((0, 0), (1, 1), (1, 1))).astype(dt)
        out = np.zeros((w.shape[0], 2 * Hh, 2 * Ww), dtype=dt)
        for a in range(2):
            for b in range(2):
                o = np.zeros((w.shape[0], Hh, Ww), dtype=dt)
                for c in range(Cc):
                    for i in range(2):
                        for j in range(2):
                            o += pw[a, b, :, c, i, j][:, None, None] * xp[c, a + i: a + i + Hh, b + j: b + j + Ww][None]
                out[:, a::2, b::2] = o
        return out

    def parity_wino22(x, w, dt):
        # F(2,2): m1 = (d0 - d1) g0, m2 = d1 (g0 + g1), m3 = (d2 - d1) g1; y0 = m1 + m2, y1 = m2 + m3
        pw = par_w(w)                                           # double
        G = np.array([[1, 0], [1, 1], [0, 1]], dtype=np.float64)
        BT = np.array([[1, -1, 0], [0, 1, 0], [0, -1, 1]], dtype=np.float64)
        AT = np.array([[1, 1, 0], [0, 1, 1]], dtype=np.float64)
        Cc, Hh, Ww = x.shape
        xp = np.pad(x, ((0, 0), (1, 1), (1, 1))).astype(dt)
        out = np.zeros((w.shape[0], 2 * Hh, 2 * Ww), dtype=dt)
        for a in range(2):
            for b in range(2):
                U = np.einsum("pi,kcij,qj->pqkc", G, pw[a, b], G).astype(dt)      # rounded once
                o = np.zeros((w.shape[0], Hh, Ww), dtype=dt)
                for ty in range(Hh // 2):
                    d = np.stack([xp[:, a + 2 * ty: a + 2 * ty + 3, b + 2 * tx: b + 2 * tx + 3] for tx in range(Ww // 2)], 0)
                    V = np.einsum("pi,tcij,qj->pqtc", BT.astype(dt), d, BT.astype(dt)).astype(dt)
                    M = np.zeros((3, 3, Ww // 2, w.shape[0]), dtype=dt)
                    for c in range(Cc):
                        M += V[:, :, :, c, None] * U[:, :, None, :, c]
                    Y = np.einsum("up,pqtk,vq->tkuv", AT.astype(dt), M, AT.astype(dt)).astype(dt)
                    for tx in range(Ww // 2):
                        o[:, 2 * ty: 2 * ty + 2, 2 * tx: 2 * tx + 2] = Y[tx]
                out[:, a::2, b::2] = o
        return out

    ref = direct3x3_up(x, w, np.float64)
    assert np.abs(parity_direct(x, w, np.float64) - ref).max() < 1e-12
    assert np.abs(parity_wino22(x, w, np.float64) - ref).max() < 1e-12
    rows = [("direct 3x3 on the up-sampled image, f32", direct3x3_up(x, w, np.float32) - ref),
            ("parity form, four 2x2-tap convolutions, f32 (shipped)", parity_direct(x, w, np.float32) - ref),
            ("parity form in F(2x2,2x2), f32", parity_wino22(x, w, np.float32) - ref)]
    base, bmax = np.sqrt((rows[0][1] ** 2).mean()), np.abs(rows[0][1]).max()
    for name, e in rows:
        print("%-56s rms %.3e  max %.3e   rms / direct %.2f   max / direct %.2f" % (name, np.sqrt((e ** 2).mean()), np.abs(e).max(), np.sqrt((e ** 2).mean()) / base, np.abs(e).max() / bmax))


if __name__ == "__main__":
    import sys
    if "--up2" in sys.argv:
        upsample_forms()
        sys.exit(0)
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
    m2 = mats([0, 1, -1])
    for name, pts in (("F(2x3,3x3) points 0, 1, -1 | 0, 1, -1, 2", [0, 1, -1, 2]), ("F(2x3,3x3) points 0, 1, -1 | 0, 1, -1, 1/2", [0, 1, -1, 0.5])):
        rows.append((name, winograd_mixed(x, w, m2, mats(pts), np.float32).astype(np.float64) - ref))
    for name, e in rows:
        print("%-42s rms %.3e  max %.3e   rms / direct %.2f   max / direct %.2f" % (
            name, np.sqrt((e ** 2).mean()), np.abs(e).max(), np.sqrt((e ** 2).mean()) / base, np.abs(e).max() / bmax))
