"""conv A -> GroupNorm+SiLU (fused statistics of A's epilogue) -> conv B with conv_bf16x3 on / off: which half differs?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ipdm_pytorch_amd import _lib, synth
from oracle import unet as ou
DEV = "cuda:0"
CASES = [(2, 128, 26, 250, 128, 3, 1, True, 2, 128), (8, 128, 64, 64, 128, 3, 1, True, 2, 256), (8, 128, 64, 64, 128, 3, 1, True, 2, 256), (8, 128, 64, 64, 128, 3, 1, False, 0, 256), (4, 128, 64, 64, 128, 3, 1, True, 2, 128), (2, 64, 80, 64, 128, 3, 1, True, 2, 128), (2, 64, 80, 64, 128, 3, 1, True, 2, 64), (2, 64, 80, 64, 64, 3, 1, True, 2, 128),
         (2, 64, 80, 64, 128, 3, 1, False, 2, 128), (2, 64, 80, 64, 128, 3, 1, True, 0, 128), (1, 128, 72, 57, 256, 3, 1, True, 2, 128)]
for case in CASES:
    B, C, H, W, CA, ksA, sA, resA, act, CB = case
    seed = 3000 + sum(case[:7])
    x = (torch.from_numpy(synth.hash_normal((B, C, H, W), seed)) * 1.3 + 0.2).to(DEV)
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (
        synth.hash_normal((CA, C, ksA, ksA), seed + 1) / np.sqrt(C * ksA * ksA), synth.hash_normal((CA,), seed + 2),
        synth.hash_uniform((CA,), seed + 5) + 0.5, synth.hash_normal((CA,), seed + 6) * 0.2,
        synth.hash_normal((CB, CA, 3, 3), seed + 3) / np.sqrt(CA * 9), synth.hash_normal((CB,), seed + 4))]
    Ho, Wo = H, W
    r = (torch.from_numpy(synth.hash_normal((B, CA, Ho, Wo), seed + 7)) * 2 - 0.7).to(DEV) if resA else None
    res = []
    for bf in (0, 1):
        d_mid = torch.full((B, CA, Ho, Wo), float("nan"), device=DEV); d_out = torch.full((B, CB, Ho, Wo), float("nan"), device=DEV)
        rows = ctypes.c_int32(-1)
        with _lib.option("conv_bf16x3", bf), _lib.option("wino2_min_tiles", 1):
            ka = _lib.lib().ipdm_conv_kernel_code_stats(B, CA, C, 3, 1, H, W); kb = _lib.lib().ipdm_conv_kernel_code(B, CB, CA, 3, 1, H, W)
            _lib.call("ipdm_op_conv_gn_conv", _lib.ptr(x), C, B, H, W, _lib.ptr(arrs[0]), _lib.ptr(arrs[1]), CA, ksA, sA, _lib.ptr(r),
                      ou.gn_groups(CA), _lib.ptr(arrs[2]), _lib.ptr(arrs[3]), act, _lib.ptr(arrs[4]), _lib.ptr(arrs[5]), CB, _lib.ptr(d_mid),
                      _lib.ptr(d_out), ctypes.byref(rows), _lib.current_stream())
        res.append((d_mid.cpu().numpy(), d_out.cpu().numpy(), ka, kb, rows.value))
    print("case %s: kernels A %d/%d B %d/%d rows %d/%d | mid max diff %.3e | out max diff %.3e (scale %.2f)" % (
        case, res[0][2], res[1][2], res[0][3], res[1][3], res[0][4], res[1][4], np.abs(res[0][0] - res[1][0]).max(), np.abs(res[0][1] - res[1][1]).max(), np.abs(res[0][1]).max()))
