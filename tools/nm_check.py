#!/usr/bin/env python
"""Narrow convolutions on the 16-cout MFMA (conv_nm.hip) against the packed-f32 VALU kernel (conv_direct.hip, option
conv_nm) and float64 torch, plus an interleaved timing A/B through the micro-benchmark entry.   python tools/nm_check.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                    # noqa: E402
import torch                          # noqa: E402
import torch.nn.functional as F       # noqa: E402
from ipdm_pytorch_amd import _lib, synth   # noqa: E402
from oracle import unet as ou         # noqa: E402

DEV = "cuda:0"
NM = int(os.environ.get("NM", "2"))        # 1: 16-cout layers only, 2: every eligible layer


def run(B, C1, C2, H, W, Cout, act, res, ks=3, seed=1):
    Cin = C1 + C2
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed))
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)) * 2 + 0.5 if C2 else None
    w = torch.from_numpy(synth.hash_normal((Cout, Cin, ks, ks), seed + 2)) / np.sqrt(Cin * ks * ks)
    bias = torch.from_numpy(synth.hash_normal((Cout,), seed + 3))
    gamma = torch.from_numpy(synth.hash_uniform((Cin,), seed + 4)) + 0.5
    beta = torch.from_numpy(synth.hash_normal((Cin,), seed + 5)) * 0.2
    groups = ou.gn_groups(Cin)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    r = torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)) if res else None

    def ref(dt):
        h = xin.to(dt)
        if act:
            h = F.group_norm(h, groups, gamma.to(dt), beta.to(dt), eps=1e-5)
            if act == 2:
                h = F.silu(h)
        o = F.conv2d(h, w.to(dt), bias.to(dt), padding=ks // 2)
        return o + r.to(dt) if res else o
    w64 = ref(torch.float64)
    w32 = ref(torch.float32)
    outs = {}
    x1d, x2d, rd = x1.to(DEV), (x2.to(DEV) if C2 else None), (r.to(DEV) if res else None)      # (kept alive across the calls)
    wn, bn, gn_, ben = (np.ascontiguousarray(t.numpy()) for t in (w, bias, gamma, beta))
    for mode in (0, 1):
        out = torch.full((B, Cout, H, W), float("nan"), device=DEV)
        with _lib.option("conv_nm", 0 if mode else NM):
            _lib.call("ipdm_op_conv2d", _lib.ptr(x1d), C1, _lib.ptr(x2d), C2, B, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, ks, 1,
                      act, groups, _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
        torch.cuda.synchronize()
        outs[mode] = out.cpu()

    def d(a):
        e = (a.double() - w64).abs()
        return float(e.max()), float((e ** 2).mean().sqrt())
    sc = float(w64.abs().max())
    (wm, wr), (dm, dr), (tm, tr) = d(outs[0]), d(outs[1]), d(w32)
    bad = int(torch.isnan(outs[0]).sum())
    print("k%d " % ks + "B%d %d+%d->%d @%dx%d act%d res%d | mfma max %.2e rms %.2e | direct max %.2e rms %.2e | torch32 max %.2e rms %.2e | "
          "mfma/direct rms %.2f  rel-max %.1e nan %d" % (B, C1, C2, Cout, H, W, act, int(res), wm, wr, dm, dr, tm, tr, wr / max(dr, 1e-30),
                                                       wm / sc, bad), flush=True)
    return wm / sc




def bench(B, C1, C2, H, W, Cout, ks, act, res, iters=20):
    ms = {}
    for rep in range(2):
        for mode in (0, 1):
            t = C.c_float()
            with _lib.option("conv_nm", 0 if mode else NM):
                _lib.call("ipdm_bench_conv2d", B, C1, C2, H, W, Cout, ks, 1, act, int(res), iters, C.byref(t))
            ms.setdefault(mode, []).append(t.value)
    by = 4.0 * B * H * W * (C1 + C2 + Cout * (2 if res else 1))
    a, b = min(ms[0]), min(ms[1])
    print("bench k%d B%d %d+%d->%d @%dx%d act%d res%d: mfma %.3f ms (%.2f TB/s algorithmic) valu %.3f ms (%.2f TB/s)  speedup %.2fx" % (
        ks, B, C1, C2, Cout, H, W, act, int(res), a, by / a / 1e9, b, by / b / 1e9, b / a), flush=True)


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    run(1, 8, 0, 16, 64, 8, 0, False)
    run(1, 8, 0, 16, 64, 8, 2, True)
    run(2, 16, 0, 37, 45, 16, 2, True)            # ragged both ways, odd width (element-wise stores)
    run(1, 16, 0, 50, 200, 16, 2, True)           # several strips, W % 64 != 0, W % 4 == 0
    run(2, 8, 4, 33, 132, 8, 2, False)            # concat 8 + 4
    run(1, 16, 16, 70, 128, 16, 2, False)         # concat, G = 8
    run(1, 16, 8, 64, 192, 8, 1, False)           # G = 6, 8 couts
    run(1, 4, 0, 40, 72, 8, 2, False)             # G = 1
    run(1, 8, 0, 20, 68, 16, 2, False)            # 8 -> 16
    run(1, 16, 0, 130, 100, 16, 0, False)
    for ks1 in ((8, 4, 8), (16, 8, 8), (8, 8, 8), (16, 16, 16), (8, 0, 16), (4, 0, 8)):
        run(2, ks1[0], ks1[1], 35, 140, ks1[2], 0, False, ks=1)
    run(1, 16, 0, 47, 61, 16, 0, True, ks=1)
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        return
    bench(8, 8, 0, 2000, 912, 8, 3, 2, True)
    bench(8, 16, 0, 1000, 456, 16, 3, 2, True)
    bench(8, 8, 0, 2000, 912, 8, 3, 2, False)
    bench(8, 16, 0, 1000, 456, 16, 3, 2, False)
    bench(8, 8, 8, 2000, 912, 8, 3, 2, False)
    bench(8, 8, 8, 2000, 912, 8, 1, 0, False)
    bench(8, 16, 16, 1000, 456, 16, 3, 2, False)
    bench(8, 16, 16, 1000, 456, 16, 1, 0, False)
    bench(8, 16, 0, 2000, 912, 16, 3, 0, False)
    bench(8, 8, 4, 2000, 912, 8, 3, 2, False)
    bench(8, 4, 0, 2000, 912, 8, 3, 2, False)
    bench(1, 8, 0, 2000, 912, 8, 3, 2, True)
    bench(1, 16, 0, 1000, 456, 16, 3, 2, True)


if __name__ == "__main__":
    main()
