"""SHA-256 of conv_wino3's outputs on a few shapes (option conv_bf16x3), for comparing two builds of the library bit for bit:
   IPDM_LIB_PATH=<lib a> python tools/experiments/w3_bits.py > a.txt; IPDM_LIB_PATH=<lib b> python tools/experiments/w3_bits.py > b.txt; diff a.txt b.txt"""
import hashlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ipdm_pytorch_amd import _lib, synth
from oracle import unet as ou
DEV = "cuda:0"
_lib.set_option("conv_bf16x3", 1)
for (B, C1, C2, H, W, Cout, act, res) in [(2, 128, 0, 228, 500, 128, 2, True), (2, 128, 16, 250, 114, 128, 2, False), (8, 256, 0, 64, 64, 256, 2, True),
                                          (3, 128, 128, 125, 57, 128, 1, False), (1, 128, 0, 130, 250, 128, 0, True), (2, 64, 64, 45, 95, 128, 1, True)]:
    Cin, seed = C1 + C2, 31 + C1 + C2 + H
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed)).to(DEV)
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)).to(DEV) if C2 else None
    rd = torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)).to(DEV) if res else None
    wn, bn, gn_, ben = (np.ascontiguousarray(t, dtype=np.float32) for t in (
        synth.hash_normal((Cout, Cin, 3, 3), seed + 2) / np.sqrt(Cin * 9), synth.hash_normal((Cout,), seed + 3),
        synth.hash_uniform((Cin,), seed + 4) + 0.5, synth.hash_normal((Cin,), seed + 5) * 0.2))
    out = torch.full((B, Cout, H, W), float("nan"), device=DEV)
    assert _lib.lib().ipdm_conv_kernel_code(B, Cout, Cin, 3, 1, H, W) == 12
    _lib.call("ipdm_op_conv2d", _lib.ptr(x1), C1, _lib.ptr(x2), C2, B, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1,
              act, ou.gn_groups(Cin), _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
    print((B, C1, C2, H, W, Cout, act, res), hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:24])
