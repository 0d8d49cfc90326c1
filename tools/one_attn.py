"""Runs one attention configuration a few times (for rocprofv3 --pmc passes): python tools/one_attn.py B heads T"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import _lib
torch.zeros(1, device="cuda")
B, heads, T = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (8, 4, 7125)
ms = C.c_float()
_lib.call("ipdm_bench_attention", B, heads, 64, T, 3, C.byref(ms))
print(B, heads, T, ms.value)
