"""Conv kernel experiment matrix (GPU box): python tools/bench_conv_dbg.py"""
import ctypes as C, sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, ipdm_pytorch_amd
    from ipdm_pytorch_amd import _lib
    torch.zeros(1, device="cuda")
    ms = C.c_float()
    for c in [(8, 64, 0, 512, 512, 64, 3, 1, 2, 1), (8, 128, 0, 512, 512, 128, 3, 1, 2, 1), (8, 128, 0, 500, 228, 128, 3, 1, 2, 1),
              (8, 128, 0, 512, 512, 128, 3, 1, 0, 0)]:
        _lib.call("ipdm_bench_conv2d", *c, 10, C.byref(ms))
        B, C1, C2, H, W, Co, ks, st, act, res = c
        print("  %-40s %7.3f ms %6.1f TF/s" % (c, ms.value, 2.0 * B * H * W * Co * (C1 + C2) * ks * ks / ms.value / 1e9))
else:
    for dbg in sys.argv[1:] or ["8"]:
        print("IPDM_CONV_DBG=%s" % dbg, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, IPDM_CONV_DBG=dbg))
