"""What exactly is wrong in the first wrong convolution of a process's first forward (conv_wino3, multiply-first wave order)?
Reads the dump of libipdm_hip_trace3.so (IPDM_TRACE_DUMP_DIR: both forwards' outputs of that convolution, its input, GroupNorm table and
Winograd-domain weights), recomputes the Winograd-domain operands of the affected workgroup tile in float64 and solves for the error
of V (the transformed activation) that explains the difference of the two outputs: which position, which channels, and what the wrong
value is relative to the right one.   usage: analyze_trace3.py <dump dir>"""
import sys
import numpy as np

d = sys.argv[1]
B, C1, Cout, H, W, idx = (int(v) for v in open(d + "/meta.txt").read().split())
out1 = np.fromfile(d + "/out1.bin", np.float32).reshape(B, Cout, H, W).astype(np.float64)
out2 = np.fromfile(d + "/out2.bin", np.float32).reshape(B, Cout, H, W).astype(np.float64)
x1 = np.fromfile(d + "/x1.bin", np.float32).reshape(B, C1, H, W).astype(np.float64)
sc = np.fromfile(d + "/sc.bin", np.float32).reshape(B, C1).astype(np.float64)
sh = np.fromfile(d + "/sh.bin", np.float32).reshape(B, C1).astype(np.float64)
u = np.fromfile(d + "/u.bin", np.float32).astype(np.float64)
nq, nct = C1 // 8, Cout // 64
# [chunk q][cout tile 64][xi 16][hh 2][lk 2][cout 32][kp 4]; channel = 8 q + 2 kp + lk, cout = 64 ct + 32 hh + cl
u = u.reshape(nq, nct, 16, 2, 2, 32, 4)
U = np.zeros((16, Cout, C1))
for q in range(nq):
    for kp in range(4):
        for lk in range(2):
            U[:, :, 8 * q + 2 * kp + lk] = u[q, :, :, :, lk, :, kp].transpose(1, 0, 2, 3).reshape(16, Cout)      # [xi][ct][hh][cl] -> cout
bad = np.argwhere(out1 != out2)
n = int(bad[0][0])
co0 = int(bad[:, 1].min()) // 128 * 128
y0 = int(bad[:, 2].min()) // 4 * 4
x0 = int(bad[:, 3].min()) // 32 * 32
rows = sorted(set(int(v) for v in bad[:, 2]))
print("convolution #%d: %d -> %d @%dx%d; %d elements differ: sample %d, couts %d..%d, rows %s (tile rows %d..%d), columns %d..%d" % (
    idx, C1, Cout, H, W, len(bad), n, bad[:, 1].min(), bad[:, 1].max(), rows, y0, y0 + 3, bad[:, 3].min(), bad[:, 3].max()))
h = x1[n] * sc[n][:, None, None] + sh[n][:, None, None]
h = h / (1.0 + np.exp(-h))                                   # SiLU (act = 2)
hp = np.zeros((C1, H + 2, W + 2)); hp[:, 1:-1, 1:-1] = h
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
slot = lambda i: i ^ (i >> 1)
dY = out1[n, co0:co0 + 128] - out2[n, co0:co0 + 128]
for ty in (0, 1):
    oy = y0 + 2 * ty
    if oy >= H or not np.any(dY[:, oy:oy + 2]):
        continue
    for tx in range(16):
        ox = x0 + 2 * tx
        if ox >= W:
            continue
        t = dY[:, oy:oy + 2, ox:ox + 2]
        if t.shape != (128, 2, 2) or not np.any(t):
            continue
        patch = hp[:, oy:oy + 4, ox:ox + 4]                 # (padded coordinates: output (oy, ox) reads input rows oy-1 .. oy+2)
        V = np.einsum("ir,crs,js->cij", BT, patch, BT)       # [c][i][j]
        # which single (i, j) explains the 2x2 differences of every cout?  dY = AT[:, i] (x) AT[:, j] * dM
        best = None
        for i in range(4):
            for j in range(4):
                basis = np.outer(AT[:, i], AT[:, j]).reshape(4)
                if not basis.any():
                    continue
                dM = t.reshape(128, 4) @ basis / (basis @ basis)
                res = np.linalg.norm(t.reshape(128, 4) - np.outer(dM, basis)) / np.linalg.norm(t)
                if best is None or res < best[0]:
                    best = (res, i, j, dM)
        res, i, j, dM = best
        Uij = U[slot(i) * 4 + j, co0:co0 + 128]             # [cout 128][cin]
        # the channels: single channels first, then the two channels of a wave (2 w, 2 w + 1 of a 16-channel chunk)
        fits = []
        for c in range(C1):
            a = Uij[:, c] @ dM / (Uij[:, c] @ Uij[:, c])
            fits.append((np.linalg.norm(dM - a * Uij[:, c]) / np.linalg.norm(dM), (c,), (a,)))
        for c in range(0, C1, 2):
            sol, *_ = np.linalg.lstsq(Uij[:, c:c + 2], dM, rcond=None)
            fits.append((np.linalg.norm(dM - Uij[:, c:c + 2] @ sol) / np.linalg.norm(dM), (c, c + 1), tuple(sol)))
        fits.sort(key=lambda f: f[0])
        f = fits[0]
        msg = "  2x2 tile (ty %d, tx %2d): position (i %d, j %d) explains it to %.1e; channels %s (chunk %d, wave %d) to %.1e:" % (
            ty, tx, i, j, res, f[1], f[1][0] // 16, (f[1][0] % 16) // 2, f[0])
        for c, a in zip(f[1], f[2]):
            v = V[c, i, j]
            b1 = np.float32(v).view(np.uint32) & np.uint32(0xffff0000)
            t1 = float(np.array(b1, np.uint32).view(np.float32))
            msg += "  c%d: dV %+.5f, V %+.5f (dV/V %+.3f; top bf16 term %+.5f)" % (c, a, v, a / v if v else float("nan"), t1)
        print(msg)
        # where does the wrong value come from?  W = V + dV against (a) the same position of every OTHER channel of this tile (stale stage content: the
        # stage held chunk s - 2 before), whole value or top term only; (b) the other fifteen positions of the same channel
        def top(v):
            return float(np.array(np.float32(v).view(np.uint32) & np.uint32(0xffff0000), np.uint32).view(np.float32))
        for c, a in zip(f[1], f[2]):
            if abs(a) < 1e-6:
                continue
            Wv = V[c, i, j] + a
            whole = sorted((abs(Wv - V[c2, i, j]), c2) for c2 in range(C1) if c2 != c)[:2]
            low = V[c, i, j] - top(V[c, i, j])
            t0 = sorted((abs((Wv - low) - top(V[c2, i, j])), c2) for c2 in range(C1) if c2 != c)[:2]
            pos = sorted((abs(Wv - V[c, i2, j2]), (i2, j2)) for i2 in range(4) for j2 in range(4) if (i2, j2) != (i, j))[:2]
            # (c) the column step of the input transform adds / subtracts two of the four row-transformed values of row i: which two give W?
            trow = np.einsum("r,rs->s", BT[i], patch[c])       # T[i][col 0..3]
            combos = sorted((abs(Wv - (sa * trow[a_] + sb * trow[b_])), "%+d*T%d %+d*T%d" % (sa, a_, sb, b_)) for a_ in range(4) for b_ in range(4) for sa in (1, -1) for sb in (1, -1, 0) if a_ != b_)[:3]
            allrows = sorted((abs(Wv - (sa * np.einsum("r,rs->s", BT[i2], patch[c])[a_] + sb * np.einsum("r,rs->s", BT[i2], patch[c])[b_])), "row %d: %+d*T%d %+d*T%d" % (i2, sa, a_, sb, b_))
                             for i2 in range(4) for a_ in range(4) for b_ in range(4) for sa in (1, -1) for sb in (1, -1) if a_ != b_)[:2]
            print("      c%d: right value = T2 - T1 of row %d (%+.6f); W as a combination of that row's T values: %s; of any row's: %s" % (
                c, i, trow[2] - trow[1], ["%s %.1e" % (n_, e) for e, n_ in combos], ["%s %.1e" % (n_, e) for e, n_ in allrows]))
            print("      c%d wrong value %+.6f: nearest same-position value of another channel %s; as [top term of another channel + own lower terms] %s; nearest other position of c%d %s" % (
                c, Wv, ["c%d %.1e" % (c2, e) for e, c2 in whole], ["c%d %.1e" % (c2, e) for e, c2 in t0], c, ["%s %.1e" % (p2, e) for e, p2 in pos]))
