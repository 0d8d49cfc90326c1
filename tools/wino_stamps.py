#!/usr/bin/env python
"""Phase stamps of conv_wino2 (diagnostic build `make -C ipdm-pytorch_amd/csrc stamps`):
   IPDM_LIB_PATH=ipdm-pytorch_amd/libipdm_hip_stamps.so python tools/wino_stamps.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ipdm_pytorch_amd import _lib   # noqa: E402

SHAPES = [(8, 128, 0, 512, 512, 128, 2, 1), (8, 128, 0, 512, 512, 128, 0, 0), (8, 64, 0, 512, 512, 64, 2, 1), (8, 256, 0, 128, 128, 256, 2, 1)]
for B, C1, C2, H, W, Co, act, res in SHAPES:
    t = C.c_float()
    chunks = (H // 4) * (W // 32) * (Co // 64) * B * ((C1 + C2) // 8) / 512.0
    print("shape B%d %d+%d->%d @%dx%d act%d res%d: %.1f chunks per workgroup" % (B, C1, C2, Co, H, W, act, res, chunks), flush=True)
    for dbg in [int(x) for x in (sys.argv[1:] or ['0', '8'])]:
        with _lib.option("conv_dbg", dbg):
            _lib.call("ipdm_bench_conv2d", B, C1, C2, H, W, Co, 3, 1, act, res, 3, C.byref(t))
        print("   conv_dbg=%d: %.3f ms" % (dbg, t.value), flush=True)
