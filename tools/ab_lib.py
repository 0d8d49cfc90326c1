"""A/B of two builds of libipdm_hip.so in ONE process, interleaved rounds (rule: perf deltas from interleaved rounds):
   python tools/ab_lib.py <libA.so> <libB.so> [<libC.so> ...]      (percentages: time relative to A, negative = faster)"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda")
libs = [C.CDLL(os.path.abspath(p)) for p in sys.argv[1:]]
for l in libs:
    l.ipdm_bench_conv2d.argtypes = [C.c_int32] * 11 + [C.POINTER(C.c_float)]
CONVS = [  # B, C1, C2, H, W, Cout, ks, stride, act, res
    (8, 64, 0, 512, 512, 64, 3, 1, 2, 1), (8, 128, 0, 256, 256, 128, 3, 1, 2, 1), (8, 64, 64, 512, 512, 64, 3, 1, 2, 0),
    (8, 128, 0, 228, 500, 128, 3, 1, 2, 1), (8, 128, 0, 512, 512, 128, 3, 1, 0, 0), (8, 256, 0, 64, 64, 256, 3, 1, 2, 1),
    (8, 256, 0, 64, 64, 768, 1, 1, 1, 0), (8, 128, 128, 228, 500, 128, 1, 1, 0, 0), (8, 128, 128, 228, 500, 128, 3, 1, 2, 0),
]
if os.environ.get("AB_NARROW"):      # the narrow layers of the projection UNet (conv_direct)
    CONVS = [(8, 8, 0, 2000, 912, 8, 3, 1, 2, 1), (8, 16, 0, 1000, 456, 16, 3, 1, 2, 1), (8, 16, 0, 2000, 912, 16, 3, 1, 0, 0),
             (8, 16, 8, 2000, 912, 8, 3, 1, 2, 0), (8, 144, 0, 1000, 456, 16, 3, 1, 2, 0), (8, 4, 0, 2000, 912, 8, 3, 1, 2, 0),
             (8, 16, 8, 2000, 912, 8, 1, 1, 0, 0), (8, 8, 0, 2000, 912, 1, 3, 1, 2, 0), (8, 1, 0, 2000, 912, 4, 3, 1, 0, 0)]
if os.environ.get("AB_WINO"):        # the layers of the 128-cout Winograd kernel (conv_wino2)
    CONVS = [(8, 128, 0, 512, 512, 128, 3, 1, 2, 1), (8, 128, 0, 512, 512, 128, 3, 1, 0, 0), (8, 256, 0, 128, 128, 256, 3, 1, 2, 1),
             (8, 128, 0, 228, 500, 128, 3, 1, 2, 1), (8, 128, 128, 228, 500, 128, 3, 1, 2, 0), (8, 256, 0, 114, 250, 256, 3, 1, 2, 1),
             (1, 128, 0, 512, 512, 128, 3, 1, 2, 1)]
if os.environ.get("AB_PW"):          # the layers of the pointwise kernel (conv_pw)
    CONVS = [(8, 256, 0, 57, 125, 768, 1, 1, 1, 0), (8, 128, 128, 228, 500, 128, 1, 1, 0, 0), (8, 256, 0, 57, 125, 256, 1, 1, 0, 1),
             (8, 256, 0, 64, 64, 768, 1, 1, 1, 0), (8, 128, 128, 256, 256, 128, 1, 1, 0, 0), (8, 256, 256, 57, 125, 256, 1, 1, 0, 0),
             (1, 128, 128, 228, 500, 128, 1, 1, 0, 0)]
if os.environ.get("AB_UP2"):         # the wide Upsample layers (conv_wup2 / the 2x2-tap parity form; H x W = the SOURCE, act 512 = Upsample mode)
    CONVS = [(8, 128, 0, 256, 256, 128, 3, 1, 512, 0), (8, 128, 0, 228, 500, 128, 3, 1, 512, 0), (8, 128, 0, 114, 250, 128, 3, 1, 512, 0),
             (8, 256, 0, 64, 64, 256, 3, 1, 512, 0), (8, 256, 0, 57, 125, 256, 3, 1, 512, 0), (1, 128, 0, 228, 500, 128, 3, 1, 512, 0),
             (1, 256, 0, 64, 64, 256, 3, 1, 512, 0)]
ATTN = [] if os.environ.get("AB_WINO") or os.environ.get("AB_PW") or os.environ.get("AB_UP2") else [(8, 4, 64, 7125), (8, 4, 64, 4096), (8, 4, 64, 1827), (8, 4, 64, 1024), (1, 4, 64, 4096)]
for l in libs:
    l.ipdm_bench_attention.argtypes = [C.c_int32] * 5 + [C.POINTER(C.c_float)]
ms = C.c_float()
res = {}
for rnd in range(2 * len(libs)):
    for c in CONVS:
        for k in range(len(libs)):            # rotate the order: the first build after a shape change runs on cold caches
            i = (k + rnd) % len(libs)
            assert libs[i].ipdm_bench_conv2d(*c, 10, C.byref(ms)) == 0
            res.setdefault((c, i), []).append(ms.value)
for c in CONVS:
    B, C1, C2, H, W, Co, ks, st, act, r = c
    fl = 2.0 * B * H * W * Co * (C1 + C2) * ks * ks
    best = [min(res[(c, i)]) for i in range(len(libs))]
    print("conv %-42s A %.3f ms %6.1f TF/s | " % (c, best[0], fl / best[0] / 1e9) +
          "  ".join("%s %+.1f%%" % (chr(66 + i - 1), 100 * (best[i] / best[0] - 1)) for i in range(1, len(libs))))
for rnd in range(2 * len(libs)):
    for c in ATTN:
        for k in range(len(libs)):
            i = (k + rnd) % len(libs)
            assert libs[i].ipdm_bench_attention(*c, 5, C.byref(ms)) == 0
            res.setdefault((c, i), []).append(ms.value)
for c in ATTN:
    B, h, d, T = c
    fl = 4.0 * B * h * T * T * d
    best = [min(res[(c, i)]) for i in range(len(libs))]
    print("attn %-42s A %.3f ms %6.1f TF/s | " % (c, best[0], fl / best[0] / 1e9) +
          "  ".join("%s %+.1f%%" % (chr(66 + i - 1), 100 * (best[i] / best[0] - 1)) for i in range(1, len(libs))))
