"""Where does conv_wino3 differ from conv_wino2?  python tools/experiments/dbg_wino3.py [case index]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ipdm_pytorch_amd import _lib, synth
from oracle import unet as ou
CASES = [(1, 128, 0, 72, 64, 128, 0, False), (2, 128, 0, 37, 145, 128, 2, True), (1, 128, 128, 24, 70, 256, 2, False), (2, 64, 64, 45, 95, 128, 1, True),
         (1, 256, 0, 17, 125, 256, 2, True), (1, 32, 0, 33, 97, 128, 2, False), (3, 128, 16, 40, 104, 128, 2, True), (1, 128, 0, 130, 250, 128, 2, True),
         (2, 128, 0, 36, 144, 128, 0, False), (2, 128, 0, 36, 144, 128, 2, False), (2, 128, 0, 36, 144, 128, 0, True), (2, 128, 0, 37, 145, 128, 0, False)]
DEV = "cuda:0"
for ci in ([int(a) for a in sys.argv[1:]] or range(len(CASES))):
    B, C1, C2, H, W, Cout, act, res = CASES[ci]
    seed = 8800 + sum(CASES[ci][:6]); Cin = C1 + C2
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed)).to(DEV)
    x2 = (torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)) * 2 + 0.5).to(DEV) if C2 else None
    w = np.ascontiguousarray((synth.hash_normal((Cout, Cin, 3, 3), seed + 2) / np.sqrt(Cin * 9)).astype(np.float32))
    bias = np.ascontiguousarray(synth.hash_normal((Cout,), seed + 3)); gamma = np.ascontiguousarray(synth.hash_uniform((Cin,), seed + 4) + 0.5)
    beta = np.ascontiguousarray(synth.hash_normal((Cin,), seed + 5) * 0.2).astype(np.float32)
    r = torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)).to(DEV) if res else None
    outs = []
    for bf in (0, 1, 1):
        out = torch.full((B, Cout, H, W), float("nan"), device=DEV)
        with _lib.option("conv_bf16x3", bf), _lib.option("wino2_min_tiles", 1):
            _lib.call("ipdm_op_conv2d", _lib.ptr(x1), C1, _lib.ptr(x2), C2, B, H, W, H, W, _lib.ptr(w), _lib.ptr(bias), Cout, 3, 1,
                      act, ou.gn_groups(Cin), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(r), _lib.ptr(out), _lib.current_stream())
        outs.append(out.cpu().numpy())
    e = np.abs(outs[1] - outs[0]); bad = e > 1e-4
    print("case %d %s: max |wino3 - wino2| %.3e, %d of %d elements above 1e-4, run-to-run equal %s" % (ci, CASES[ci], e.max(), bad.sum(), e.size, np.array_equal(outs[1], outs[2])))
    if bad.any():
        idx = np.argwhere(bad)
        for d, name in enumerate(("b", "cout", "y", "x")):
            vals, cnt = np.unique(idx[:, d], return_counts=True)
            print("   %s: %d distinct; first %s counts %s" % (name, len(vals), vals[:24].tolist(), cnt[:24].tolist()))
