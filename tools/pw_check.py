"""conv_pw.hip (barrier-free pointwise kernel) against conv_ws.hip's 1x1 path: outputs of both through ipdm_op_conv2d on the
same inputs (option conv_no_pw switches per call), repeated to catch run-dependent errors, then interleaved timing of the
layer shapes of the two UNets.      python tools/pw_check.py [check|bench|all]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ipdm_pytorch_amd import _lib

DEV = "cuda"
mode = sys.argv[1] if len(sys.argv) > 1 else "all"


def op_conv(x, w, b, act=0, gamma=None, beta=None, res=None, x2=None, groups=32):
    B, C1, H, W = x.shape
    C2 = 0 if x2 is None else x2.shape[1]
    Cout = w.shape[0]
    out = torch.full((B, Cout, H, W), float("nan"), device=DEV)
    arrs = [None if t is None else np.ascontiguousarray(t.cpu().numpy(), dtype=np.float32) for t in (w, b, gamma, beta)]
    _lib.call("ipdm_op_conv2d", _lib.ptr(x), C1, _lib.ptr(x2), C2, B, H, W, H, W, _lib.ptr(arrs[0]), _lib.ptr(arrs[1]),
              Cout, 1, 1, act, groups if act else 0, _lib.ptr(arrs[2]), _lib.ptr(arrs[3]), _lib.ptr(res), _lib.ptr(out),
              _lib.current_stream())
    return out


def code(B, Cout, Cin, H, W):
    return _lib.lib().ipdm_conv_kernel_code(B, Cout, Cin, 1, 1, H, W)


CHECKS = [  # B, C1, C2, H, W, Cout, act, res
    (2, 256, 0, 57, 125, 768, 1, 0), (2, 256, 0, 57, 125, 256, 0, 1), (2, 128, 128, 30, 44, 128, 0, 0), (1, 64, 0, 64, 64, 128, 0, 0),
    (3, 96, 32, 37, 41, 128, 1, 1), (2, 256, 256, 29, 63, 256, 0, 0), (1, 128, 0, 512, 512, 128, 0, 1), (8, 256, 0, 64, 64, 768, 1, 0),
]
if mode in ("check", "all"):
    g = torch.Generator().manual_seed(7)
    for c in CHECKS:
        B, C1, C2, H, W, Cout, act, res = c
        x = torch.randn((B, C1, H, W), generator=g).to(DEV) * 1.5 + 0.3
        x2 = torch.randn((B, C2, H, W), generator=g).to(DEV) if C2 else None
        w = torch.randn((Cout, C1 + C2, 1, 1), generator=g) / (C1 + C2) ** 0.5
        b = torch.randn((Cout,), generator=g)
        gamma = torch.randn((C1 + C2,), generator=g) if act else None
        beta = torch.randn((C1 + C2,), generator=g) if act else None
        r = torch.randn((B, Cout, H, W), generator=g).to(DEV) if res else None
        with _lib.option("conv_no_pw", 1):
            kc_ws = code(B, Cout, C1 + C2, H, W)
            want = op_conv(x, w, b, act, gamma, beta, r, x2)
        with _lib.option("pw_force", 1):
            kc = code(B, Cout, C1 + C2, H, W)
        worst, nbad = 0.0, 0
        for rep in range(6):
            with _lib.option("pw_item", 1 + rep % 2), _lib.option("pw_force", 1):      # (small shapes: past the launcher's fill rule)      # both item shapes: the same bits
                got = op_conv(x, w, b, act, gamma, beta, r, x2)
            d = (got - want).abs().max().item()
            worst = max(worst, d)
            nbad += int(not torch.isfinite(got).all().item())
            if rep == 0:
                first = got
            elif not torch.equal(got, first):
                nbad += 1
        print("check %-40s kernel %d (ws: %d)  max|pw - ws| %.3e  (|out| max %.2f)  bad runs %d" % (c, kc, kc_ws, worst, want.abs().max().item(), nbad))

BENCH = [  # B, C1, C2, H, W, Cout, ks, stride, act, res
    (8, 256, 0, 57, 125, 768, 1, 1, 1, 0), (8, 128, 128, 228, 500, 128, 1, 1, 0, 0), (8, 256, 0, 57, 125, 256, 1, 1, 0, 1),
    (8, 256, 0, 64, 64, 768, 1, 1, 1, 0), (8, 128, 128, 256, 256, 128, 1, 1, 0, 0), (8, 256, 0, 64, 64, 256, 1, 1, 0, 1),
    (8, 256, 0, 29, 63, 768, 1, 1, 1, 0), (8, 256, 128, 114, 250, 128, 1, 1, 0, 0), (8, 256, 256, 57, 125, 256, 1, 1, 0, 0),
    (1, 256, 0, 57, 125, 768, 1, 1, 1, 0), (1, 128, 128, 228, 500, 128, 1, 1, 0, 0), (1, 256, 0, 64, 64, 768, 1, 1, 1, 0),
]
if mode in ("bench", "all"):
    ms = C.c_float()
    res = {}
    ARMS = [("conv_no_pw", 0, "pw_item", 0, 0), ("conv_no_pw", 1, "pw_item", 0, 0), ("conv_no_pw", 0, "pw_item", 1, 1), ("conv_no_pw", 0, "pw_item", 2, 1)]      # (the forced item shapes also force the kernel)
    for rnd in range(8):
        for c in BENCH:
            for k in range(4):
                i = (k + rnd) % 4
                with _lib.option(ARMS[i][0], ARMS[i][1]), _lib.option(ARMS[i][2], ARMS[i][3]), _lib.option("pw_force", ARMS[i][4]):
                    _lib.call("ipdm_bench_conv2d", *c, 10, C.byref(ms))
                res.setdefault((c, i), []).append(ms.value)
    for c in BENCH:
        B, C1, C2, H, W, Co, ks, st, act, r = c
        fl = 2.0 * B * H * W * Co * (C1 + C2)
        pw, ws, p32, p64 = (min(res[(c, i)]) for i in range(4))
        print("conv %-44s default %.3f ms %6.1f TF/s (%.2f of peak) | ws %.3f ms %6.1f TF/s | pw/ws %.2f | 32-pixel items %.2f  64-pixel items %.2f" %
              (c, pw, fl / pw / 1e9, fl / pw / 1e9 / 157.3, ws, fl / ws / 1e9, pw / ws, p32 / ws, p64 / ws))
