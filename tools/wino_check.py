#!/usr/bin/env python
"""Winograd-domain convolution (conv_wino.hip) against the direct kernel and float64 torch, plus an interleaved timing
A/B (option conv_no_wino) through the micro-benchmark entry.   python tools/wino_check.py [quick]"""
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


def run(B, C1, C2, H, W, Cout, act, res, seed=1):
    Cin = C1 + C2
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed))
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)) * 2 + 0.5 if C2 else None
    w = torch.from_numpy(synth.hash_normal((Cout, Cin, 3, 3), seed + 2)) / np.sqrt(Cin * 9)
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
        o = F.conv2d(h, w.to(dt), bias.to(dt), padding=1)
        return o + r.to(dt) if res else o
    w64 = ref(torch.float64)
    w32 = ref(torch.float32)
    outs = {}
    x1d, x2d, rd = x1.to(DEV), (x2.to(DEV) if C2 else None), (r.to(DEV) if res else None)      # (kept alive across the calls)
    wn, bn, gn_, ben = (np.ascontiguousarray(t.numpy()) for t in (w, bias, gamma, beta))
    codes = {}
    for mode in (0, 1):
        out = torch.full((B, Cout, H, W), float("nan"), device=DEV)
        with _lib.option("conv_no_wino", mode):
            codes[mode] = _lib.lib().ipdm_conv_kernel_code(B, Cout, Cin, 3, 1, H, W)      # (1 / 2: Winograd 64- / 128-cout tiles, 3 / 4: direct)
            _lib.call("ipdm_op_conv2d", _lib.ptr(x1d), C1, _lib.ptr(x2d), C2, B, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1,
                      act, groups, _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
        torch.cuda.synchronize()
        outs[mode] = out.cpu()

    def d(a):
        e = (a.double() - w64).abs()
        return float(e.max()), float((e ** 2).mean().sqrt())
    sc = float(w64.abs().max())
    (wm, wr), (dm, dr), (tm, tr) = d(outs[0]), d(outs[1]), d(w32)
    bad = int(torch.isnan(outs[0]).sum())
    print("B%d %d+%d->%d @%dx%d act%d res%d | wino max %.2e rms %.2e | direct max %.2e rms %.2e | torch32 max %.2e rms %.2e | "
          "wino/direct rms %.2f  rel-max %.1e nan %d  kernels %d/%d%s" % (B, C1, C2, Cout, H, W, act, int(res), wm, wr, dm, dr, tm, tr, wr / max(dr, 1e-30),
                                                       wm / sc, bad, codes[0], codes[1], "" if codes[0] in (1, 2) else "  (NOT Winograd: this shape falls back)"), flush=True)
    return wm / sc


def bench(B, C1, C2, H, W, Cout, act, res, iters=20, planar=False):
    act = act | (256 if planar else 0)
    ms = {}
    for rep in range(2):
        for mode in (0, 1, 2):          # 0: Winograd (round-4 kernel), 1: direct conv_ws, 2: Winograd, round-3 kernel
            t = C.c_float()
            with _lib.option("conv_no_wino", int(mode == 1)), _lib.option("wino_v1", int(mode == 2)):
                _lib.call("ipdm_bench_conv2d", B, C1, C2, H, W, Cout, 3, 1, act, int(res), iters, C.byref(t))
            ms.setdefault(mode, []).append(t.value)
    fl = 2.0 * B * H * W * Cout * (C1 + C2) * 9
    a, b, v1 = min(ms[0]), min(ms[1]), min(ms[2])
    act &= 255
    print(("planar " if planar else "") + "bench B%d %d+%d->%d @%dx%d act%d res%d: wino %.3f ms (%.1f TF/s-equivalent, %.3f of the f32 peak executed) "
          "r03 kernel %.3f ms (%.2fx) direct %.3f ms (%.1f TF/s)  speedup %.2fx" % (
        B, C1, C2, Cout, H, W, act, int(res), a, fl / a / 1e9, fl * 16 / 36 / a / 1e9 / 157.3, v1, v1 / a, b, fl / b / 1e9, b / a), flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    run(1, 64, 0, 8, 32, 64, 0, False)
    run(1, 64, 0, 8, 32, 64, 2, True)
    run(2, 64, 0, 37, 45, 64, 2, True)           # ragged both ways, odd width
    run(1, 128, 64, 16, 32, 128, 2, False)       # concat
    run(2, 128, 0, 64, 64, 128, 2, True)
    run(1, 256, 0, 32, 57, 256, 1, False)        # K-split shape stays direct (few tiles): both rows equal
    run(1, 64, 0, 130, 250, 128, 2, True)
    run(1, 128, 0, 96, 64, 64, 2, True)
    run(2, 64, 64, 70, 131, 64, 1, True)         # concat, odd width
    run(1, 64, 0, 64, 96, 64, 0, False)
    run(1, 64, 0, 76, 65, 64, 2, False)          # ox0 + 33 == W: the last part of an interior tile ends with the row
    run(1, 64, 0, 75, 40, 64, 2, True)           # right-edge parts straddle the row end; odd height
    run(1, 64, 0, 33, 97, 64, 0, False)
    run(3, 64, 0, 50, 70, 128, 1, False)
    run(2, 32, 0, 61, 129, 64, 2, True)          # 4 chunks (the minimum), odd x odd
    if os.environ.get("IPDM_LIB_PATH"):          # stamps build: phase cycles of two shapes
        with _lib.option("conv_dbg", int(os.environ.get("WINO_DBG", "8"))):
            bench(8, 128, 0, 512, 512, 128, 2, True, iters=3)
            bench(8, 128, 0, 512, 512, 128, 0, False, iters=3)
            bench(8, 64, 0, 512, 512, 64, 2, True, iters=3)
            for pl in (False, True):             # the same concat layer from an NCHW / a parity-planar x1
                bench(8, 64, 64, 512, 512, 64, 2, False, iters=3, planar=pl)
        return
    bench(8, 128, 0, 512, 512, 128, 2, True)
    if quick:
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ksplit":      # the K-split layers: K slices inside conv_wino2 vs the K-split direct kernel
        for B in (1, 8):
            for shp in ((256, 0, 32, 32), (256, 0, 63, 29), (256, 256, 63, 29), (256, 256, 32, 32), (256, 128, 63, 29)):
                bench(B, shp[0], shp[1], shp[2], shp[3], 256, 2, shp[1] == 0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        for B in (1, 2, 4, 8):
            for shp in ((128, 128, 128), (256, 128, 128), (256, 64, 64), (128, 250, 114), (256, 125, 57), (256, 250, 114)):
                bench(B, shp[0], 0, shp[1], shp[2], shp[0], 2, True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "planar":
        for pl in (False, True):
            bench(8, 64, 64, 512, 512, 64, 2, False, planar=pl)
            bench(8, 128, 64, 256, 256, 64, 2, False, planar=pl)
            bench(8, 128, 128, 128, 128, 128, 2, False, planar=pl)
            bench(8, 128, 128, 228, 500, 128, 2, False, planar=pl)
        return
    bench(8, 64, 0, 512, 512, 64, 2, True)
    bench(8, 128, 0, 512, 512, 128, 0, False)
    bench(8, 256, 0, 128, 128, 256, 2, True)
    bench(8, 128, 0, 228, 500, 128, 2, True)
    bench(8, 256, 0, 114, 250, 256, 2, True)
    bench(8, 256, 128, 114, 250, 256, 2, False)
    bench(8, 256, 0, 64, 64, 256, 2, True)
    bench(8, 128, 0, 128, 128, 128, 2, True)
    bench(8, 256, 0, 125, 57, 256, 2, True)
    bench(1, 128, 0, 512, 512, 128, 2, True)
    bench(1, 128, 0, 256, 256, 128, 2, True)
    bench(1, 128, 0, 128, 128, 128, 2, True)
    bench(1, 256, 0, 128, 128, 256, 2, True)
    bench(1, 256, 0, 64, 64, 256, 2, True)
    bench(1, 128, 0, 500, 228, 128, 2, True)
    bench(1, 128, 0, 250, 114, 128, 2, True)
    bench(1, 256, 0, 125, 57, 256, 2, True)


if __name__ == "__main__":
    main()
