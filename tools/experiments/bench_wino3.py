"""conv_wino3 (option conv_bf16x3) against conv_wino2 on the dominant kernel's shapes, interleaved rounds in one process:
   python tools/experiments/bench_wino3.py <lib.so> [<variant lib.so> ...]   (first lib: conv_wino2 AND conv_wino3; others: conv_wino3 only)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.zeros(1, device="cuda")
libs = [C.CDLL(os.path.abspath(p)) for p in sys.argv[1:]]
arms = [(libs[0], 0, "wino2")] + [(l, 1, os.path.basename(p).replace("libipdm_hip", "w3").replace(".so", "")) for l, p in zip(libs, sys.argv[1:])]
for l in libs:
    l.ipdm_bench_conv2d.argtypes = [C.c_int32] * 11 + [C.POINTER(C.c_float)]
    l.ipdm_set_option.argtypes = [C.c_char_p, C.c_int]
CONVS = [(8, 128, 0, 512, 512, 128, 3, 1, 2, 1), (8, 128, 0, 512, 512, 128, 3, 1, 0, 0), (8, 256, 0, 128, 128, 256, 3, 1, 2, 1),
         (8, 128, 0, 228, 500, 128, 3, 1, 2, 1), (8, 128, 128, 228, 500, 128, 3, 1, 2, 0), (8, 256, 0, 114, 250, 256, 3, 1, 2, 1),
         (1, 128, 0, 512, 512, 128, 3, 1, 2, 1)]
ms = C.c_float()
res = {}
for rnd in range(2 * len(arms)):
    for c in CONVS:
        for k in range(len(arms)):
            i = (k + rnd) % len(arms)
            lib, bf, _ = arms[i]
            assert lib.ipdm_set_option(b"conv_bf16x3", bf) == 0
            assert lib.ipdm_bench_conv2d(*c, 10, C.byref(ms)) == 0
            res.setdefault((c, i), []).append(ms.value)
for c in CONVS:
    B, C1, C2, H, W, Co, ks, st, act, r = c
    fl = 2.0 * B * H * W * Co * (C1 + C2) * 9
    best = [min(res[(c, i)]) for i in range(len(arms))]
    print("conv %-42s wino2 %.3f ms %6.1f TF/s (3x3 form) | " % (c, best[0], fl / best[0] / 1e9) +
          "  ".join("%s %.3f ms x%.2f" % (arms[i][2], best[i], best[0] / best[i]) for i in range(1, len(arms))))
