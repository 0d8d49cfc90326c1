"""What reading x1 parity-planar costs the Winograd kernels: the same layer timed with x1 as NCHW and as an Upsample's parity-planar
output (the micro-benchmark's act bit 256).   python tools/experiments/planar_penalty.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import _lib
torch.zeros(1, device="cuda")
ms = C.c_float()
for shape in [(8, 128, 128, 256, 256, 128), (8, 128, 64, 512, 512, 64), (8, 256, 128, 128, 128, 128), (8, 256, 256, 64, 64, 256),
              (8, 128, 128, 228, 500, 128), (8, 256, 128, 114, 250, 128), (1, 128, 64, 512, 512, 64), (1, 128, 128, 256, 256, 128)]:
    B, C1, C2, H, W, Co = shape
    t = []
    for fl in (0, 256):
        best = 1e30
        for _ in range(3):
            _lib.call("ipdm_bench_conv2d", B, C1, C2, H, W, Co, 3, 1, 2 | fl, 0, 10, C.byref(ms))
            best = min(best, ms.value)
        t.append(best)
    print("%-36s NCHW %.3f ms   parity-planar x1 %.3f ms   %+.1f %%" % (shape, t[0], t[1], 100 * (t[1] / t[0] - 1)), flush=True)
