"""Times the wide Upsample layers of the two reference networks in both parity forms: the 2x2-tap convolutions (conv_ws.hip,
option conv_no_wup2 = 1) and their Winograd F(2x2,2x2) form (conv_wup2.hip).   python tools/up2_bench.py [B ...]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import _lib

LAYERS = [("img 128 @256x256 -> 512x512", 128, 256, 256), ("img 128 @128x128", 128, 128, 128), ("img 256 @64x64", 256, 64, 64),
          ("img 256 @32x32", 256, 32, 32), ("img 256 @16x16", 256, 16, 16),
          ("proj 128 @228x500 -> 456x1000", 128, 228, 500), ("proj 128 @114x250", 128, 114, 250), ("proj 256 @57x125", 256, 57, 125),
          ("proj 128 @500x228 (not transposed)", 128, 500, 228)]
torch.zeros(1, device="cuda")
ms = C.c_float()
for B in [int(v) for v in sys.argv[1:]] or [8, 1]:
    tot = [0.0, 0.0]
    for name, ch, H, W in LAYERS:
        t = []
        for off in (1, 0):
            with _lib.option("conv_no_wup2", off):
                best = 1e30
                for _ in range(2):
                    _lib.call("ipdm_bench_conv2d", B, ch, 0, H, W, ch, 3, 1, 512, 0, 10, C.byref(ms))
                    best = min(best, ms.value)
                t.append(best)
        gf = 2.0 * B * H * W * ch * ch / 1e9
        print("B=%d  %-40s  2x2-tap %7.3f ms (%5.1f TF/s of 16)   F(2x2,2x2) %7.3f ms (%5.1f TF/s of 9)   x%.2f" % (
            B, name, t[0], gf * 16 / t[0], t[1], gf * 9 / t[1], t[0] / t[1]), flush=True)
        if "not transposed" not in name:
            tot[0] += t[0]; tot[1] += t[1]
    print("B=%d  sum %.3f -> %.3f ms" % (B, tot[0], tot[1]))
