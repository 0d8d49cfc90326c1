"""Runs one conv configuration a few times (for rocprofv3 --pmc passes): python tools_one_conv.py B C1 C2 H W Cout ks stride act res"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import _lib
torch.zeros(1, device="cuda")
cfg = [int(v) for v in sys.argv[1:11]] if len(sys.argv) > 10 else [8, 128, 0, 512, 512, 128, 3, 1, 0, 0]
ms = C.c_float()
_lib.call("ipdm_bench_conv2d", *cfg, 3, C.byref(ms))
print(cfg, ms.value)
