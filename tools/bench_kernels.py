"""Kernel micro-benchmarks on the GPU box: python tools_bench_kernels.py  (tuning aid)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import _lib

torch.cuda.init(); torch.zeros(1, device="cuda")
ms = C.c_float()
CONVS = [  # B, C1, C2, H, W, Cout, ks, stride, act, res
    (8, 64, 0, 512, 512, 64, 3, 1, 2, 1), (8, 128, 0, 512, 512, 128, 3, 1, 0, 0), (8, 128, 64, 512, 512, 64, 3, 1, 2, 0),
    (8, 128, 0, 256, 256, 128, 3, 1, 2, 1), (8, 256, 0, 64, 64, 256, 3, 1, 2, 1), (8, 256, 0, 32, 32, 256, 3, 1, 2, 1),
    (8, 128, 0, 500, 228, 128, 3, 1, 2, 1), (8, 8, 0, 2000, 912, 8, 3, 1, 2, 1), (8, 16, 0, 1000, 456, 16, 3, 1, 2, 1),
    (8, 256, 0, 64, 64, 768, 1, 1, 1, 0), (8, 256, 0, 64, 64, 256, 1, 1, 0, 1), (8, 128, 64, 512, 512, 64, 1, 1, 0, 0),
    (8, 64, 0, 512, 512, 64, 3, 2, 0, 0),
]
for c in CONVS:
    _lib.call("ipdm_bench_conv2d", *c, 10, C.byref(ms))
    B, C1, C2, H, W, Co, ks, st, act, res = c
    Ho, Wo = (H + 2 * (ks // 2) - ks) // st + 1, (W + 2 * (ks // 2) - ks) // st + 1
    fl = 2.0 * B * Ho * Wo * Co * (C1 + C2) * ks * ks
    print("conv %-44s %8.3f ms  %7.1f TF/s" % (c, ms.value, fl / ms.value / 1e9))
for (B, h, d, T) in ((8, 4, 64, 4096), (8, 4, 64, 1024), (8, 4, 64, 7125), (8, 4, 64, 1827)):
    _lib.call("ipdm_bench_attention", B, h, d, T, 5, C.byref(ms))
    print("attn B=%d T=%-5d %8.3f ms  %7.1f TF/s" % (B, T, ms.value, 4.0 * B * h * T * T * d / ms.value / 1e9))
