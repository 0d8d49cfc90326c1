"""Option conv_bf16x3: is a batch its slices, layer by layer?  Every wide 3x3 shape of the two production UNets, B = 2 and 3 against
the slices run alone; prints the kernel codes (default / option, lone slice / batch) and the largest difference."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ipdm_pytorch_amd import _lib, synth
from oracle import unet as ou
DEV = "cuda:0"
code = _lib.lib().ipdm_conv_kernel_code
shapes = []
for (H, W) in [(250, 114), (125, 57), (63, 29), (32, 15), (128, 128), (64, 64), (32, 32), (16, 16), (256, 256)]:
    for (C1, C2, Cout) in [(128, 0, 128), (128, 16, 128), (128, 128, 128), (128, 0, 256), (256, 0, 256), (256, 256, 256), (256, 128, 128), (256, 128, 256), (64, 64, 128)]:
        shapes.append((C1, C2, H, W, Cout))
bad = 0
for (C1, C2, H, W, Cout) in shapes:
    for B in (2, 3):
        for act, res in ((2, True), (0, False)):
            Cin, seed = C1 + C2, 77 + C1 + C2 + H
            x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed)).to(DEV)
            x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)).to(DEV) if C2 else None
            rd = torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)).to(DEV) if res else None
            wn, bn, gn_, ben = (np.ascontiguousarray(t, dtype=np.float32) for t in (
                synth.hash_normal((Cout, Cin, 3, 3), seed + 2) / np.sqrt(Cin * 9), synth.hash_normal((Cout,), seed + 3),
                synth.hash_uniform((Cin,), seed + 4) + 0.5, synth.hash_normal((Cin,), seed + 5) * 0.2))

            def run(lo, hi):
                out = torch.full((hi - lo, Cout, H, W), float("nan"), device=DEV)
                a, b, r = x1[lo:hi].contiguous(), (x2[lo:hi].contiguous() if C2 else None), (rd[lo:hi].contiguous() if res else None)
                _lib.call("ipdm_op_conv2d", _lib.ptr(a), C1, _lib.ptr(b), C2, hi - lo, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1,
                          act, ou.gn_groups(Cin), _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(r), _lib.ptr(out), _lib.current_stream())
                return out
            d0 = (code(1, Cout, Cin, 3, 1, H, W), code(B, Cout, Cin, 3, 1, H, W))
            with _lib.option("conv_bf16x3", 1):
                d1 = (code(1, Cout, Cin, 3, 1, H, W), code(B, Cout, Cin, 3, 1, H, W))
                whole = run(0, B)
                diff = max((run(i, i + 1)[0] - whole[i]).abs().max().item() for i in range(B))
            if diff or d1[0] != d1[1]:
                bad += 1
            print("%4d+%-3d -> %3d @%3dx%-3d B=%d act=%d res=%d codes default %s option %s  batch-vs-slices %.2e%s" % (
                C1, C2, Cout, H, W, B, act, int(res), d0, d1, diff, "   <<<" if diff else ""), flush=True)
print("layers where a batch is not its slices:", bad)
