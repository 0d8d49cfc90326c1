"""Direct narrow conv experiment matrix (GPU box): python tools/bench_direct_dbg.py"""
import ctypes as C, sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch, ipdm_pytorch_amd
    from ipdm_pytorch_amd import _lib
    torch.zeros(1, device="cuda")
    ms = C.c_float()
    for c in [(8, 8, 0, 2000, 912, 8, 3, 1, 2, 1), (8, 16, 0, 1000, 456, 16, 3, 1, 2, 1), (8, 16, 8, 2000, 912, 8, 3, 1, 2, 0), (8, 128, 16, 1000, 456, 16, 3, 1, 2, 0)]:
        _lib.call("ipdm_bench_conv2d", *c, 10, C.byref(ms))
        B, C1, C2, H, W, Co, ks, st, act, res = c
        gb = B * H * W * 4 * ((C1 + C2) + Co * (2 if res else 1)) / 1e9
        print("  %-40s %7.3f ms %6.1f TF/s  %5.2f TB/s" % (c, ms.value, 2.0 * B * H * W * Co * (C1 + C2) * ks * ks / ms.value / 1e9, gb / ms.value))
else:
    for dbg in sys.argv[1:] or ["0", "1", "2", "3", "4", "7"]:
        print("IPDM_CONV_DBG=%s" % dbg, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, IPDM_CONV_DBG=dbg))
