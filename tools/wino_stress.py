#!/usr/bin/env python
"""Run-to-run determinism of conv_wino2 under load: every shape 40 times, outputs compared bit for bit with the first run and
with the 64-cout kernel (wino_v1).  Includes launches with ONE tile per workgroup (waves end right behind their last stores),
K-split launches and many-round launches.   python tools/wino_stress.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                    # noqa: E402
import torch                          # noqa: E402
from ipdm_pytorch_amd import _lib, synth   # noqa: E402
from oracle import unet as ou         # noqa: E402

DEV = "cuda:0"


def once(B, C1, C2, H, W, Cout, act, res, tensors):
    x1d, x2d, rd, wn, bn, gn_, ben, groups = tensors
    out = torch.full((B, Cout, H, W), float("nan"), device=DEV)
    _lib.call("ipdm_op_conv2d", _lib.ptr(x1d), C1, _lib.ptr(x2d), C2, B, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1,
              act, groups, _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
    return out


def stress(B, C1, C2, H, W, Cout, act, res, reps=40, seed=5):
    Cin = C1 + C2
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed))
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)) if C2 else None
    w = torch.from_numpy(synth.hash_normal((Cout, Cin, 3, 3), seed + 2)) / np.sqrt(Cin * 9)
    bias = torch.from_numpy(synth.hash_normal((Cout,), seed + 3))
    gamma = torch.from_numpy(synth.hash_uniform((Cin,), seed + 4)) + 0.5
    beta = torch.from_numpy(synth.hash_normal((Cin,), seed + 5)) * 0.2
    r = torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)) if res else None
    tensors = (x1.to(DEV), x2.to(DEV) if C2 else None, r.to(DEV) if res else None) + tuple(
        np.ascontiguousarray(t.numpy()) for t in (w, bias, gamma, beta)) + (ou.gn_groups(Cin),)
    with _lib.option("wino2_min_tiles", 1):
        code = _lib.lib().ipdm_conv_kernel_code(B, Cout, Cin, 3, 1, H, W)
        first = once(B, C1, C2, H, W, Cout, act, res, tensors)
        bad = 0
        for _ in range(reps):
            o = once(B, C1, C2, H, W, Cout, act, res, tensors)
            bad += int(not torch.equal(o, first))
    ref = None
    if code == 2:
        with _lib.option("wino_v1", 1):
            ref = once(B, C1, C2, H, W, Cout, act, res, tensors)
    print("B%d %d+%d->%d @%dx%d act%d res%d kernel %d: %d of %d runs differ from the first; %s" % (
        B, C1, C2, Cout, H, W, act, int(res), code, bad, reps,
        "equal to the 64-cout kernel" if ref is not None and torch.equal(ref, first) else ("DIFFERS from the 64-cout kernel" if ref is not None else "-")), flush=True)
    return bad


if __name__ == "__main__":
    total = 0
    total += stress(1, 128, 0, 72, 64, 128, 0, False)        # 36 tiles: one per workgroup
    total += stress(1, 128, 0, 72, 64, 128, 2, True)
    total += stress(2, 128, 0, 100, 96, 256, 2, False)       # 300 tiles: one round + a tail
    total += stress(8, 128, 0, 128, 128, 128, 2, True)       # 4 rounds
    total += stress(1, 256, 0, 32, 32, 256, 2, True)         # K slices
    total += stress(8, 256, 256, 29, 63, 256, 2, False)      # K slices over a concat
    total += stress(1, 128, 16, 61, 129, 128, 1, True)       # odd sizes, 144 channels
    print("TOTAL mismatching runs:", total)
    sys.exit(1 if total else 0)
