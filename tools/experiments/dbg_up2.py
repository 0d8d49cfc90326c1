import os, sys, ctypes, torch, numpy as np
import torch.nn.functional as F
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_parity as T
from ipdm_pytorch_amd import _lib
synth, DEV = T.synth, T.DEV
def run(case):
    B, C, Hs, Ws, CA, C2, CB, ksB, act = case
    seed = 5000 + sum(case)
    x = torch.from_numpy(synth.hash_normal((B, C, Hs, Ws), seed)) * 1.1 + 0.3
    wA = torch.from_numpy(synth.hash_normal((CA, C, 3, 3), seed + 1)) / np.sqrt(C * 9)
    bA = torch.from_numpy(synth.hash_normal((CA,), seed + 2))
    wB = torch.from_numpy(synth.hash_normal((CB, CA, ksB, ksB), seed + 3)) / np.sqrt(CA * ksB * ksB)
    bB = torch.from_numpy(synth.hash_normal((CB,), seed + 4))
    gamma = torch.ones(CA); beta = torch.zeros(CA)
    mid = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wA, bA, padding=1)
    d_mid = torch.full(tuple(mid.shape), float("nan"), device=DEV)
    d_out = torch.full((B, CB, 2 * Hs, 2 * Ws), float("nan"), device=DEV)
    arrs = [np.ascontiguousarray(t.numpy(), dtype=np.float32) for t in (wA, bA, gamma, beta, wB, bB)]
    used = ctypes.c_int32(-1)
    _lib.call("ipdm_op_up_conv_chain", _lib.ptr(x.to(DEV)), C, B, Hs, Ws, _lib.ptr(arrs[0]), _lib.ptr(arrs[1]), CA, None, 0,
              T.ou.gn_groups(CA), _lib.ptr(arrs[2]), _lib.ptr(arrs[3]), act, _lib.ptr(arrs[4]), _lib.ptr(arrs[5]), CB, ksB, _lib.ptr(d_mid),
              _lib.ptr(d_out), ctypes.byref(used), _lib.current_stream())
    e = (d_mid.cpu() - mid).abs()
    bad = e > 1e-4
    print(case, "used", used.value, "max", float(e.max()), "bad count", int(bad.sum()), "nan", int(torch.isnan(d_mid.cpu()).sum()))
    if bad.any():
        idx = bad.nonzero()
        print(" channels", idx[:, 1].unique()[:40].tolist())
        print(" rows", idx[:, 2].unique()[:60].tolist())
        print(" cols", idx[:, 3].unique()[:80].tolist())
        print(" first", idx[:10].tolist())
for case in [(1, 128, 114, 250, 128, 0, 128, 3, 2), (1, 128, 8, 34, 128, 0, 128, 3, 2), (1, 128, 8, 31, 128, 0, 128, 3, 2), (1, 128, 8, 67, 128, 0, 128, 3, 2)]:
    run(case)
