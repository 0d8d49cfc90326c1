import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
import ipdm_pytorch_amd
from ipdm_pytorch_amd import _lib, synth
torch.set_num_threads(32)
def run(B, C1, C2, H, W, Cout, act, res, seed=5):
    Cin = C1 + C2
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed))
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)) if C2 else None
    w = torch.from_numpy(synth.hash_normal((Cout, Cin, 3, 3), seed + 2)) * (1.0 / (3 * Cin ** 0.5))
    b = torch.from_numpy(synth.hash_normal((Cout,), seed + 3))
    gamma = torch.from_numpy(synth.hash_normal((Cin,), seed + 4)) * 0.2 + 1
    beta = torch.from_numpy(synth.hash_normal((Cin,), seed + 5)) * 0.2
    h = x1 if x2 is None else torch.cat([x1, x2], 1)
    hd = h.double()
    groups = 32 if Cin % 32 == 0 else (Cin if Cin < 32 else 4)
    if act:
        hd = F.group_norm(hd, groups, gamma.double(), beta.double(), eps=1e-5)
        if act == 2: hd = F.silu(hd)
    want = F.conv2d(hd, w.double(), b.double(), padding=1)
    r = torch.from_numpy(synth.hash_normal(tuple(want.shape), seed + 6)) if res else None
    if res: want = want + r.double()
    out = torch.empty(tuple(want.shape), device="cuda")
    args = [np.ascontiguousarray(t.numpy()) for t in (w, b, gamma, beta)]
    _lib.call("ipdm_op_conv2d", _lib.ptr(x1.cuda()), C1, _lib.ptr(x2.cuda()) if C2 else None, C2, B, H, W, H, W, _lib.ptr(args[0]), _lib.ptr(args[1]),
              Cout, 3, 1, act, groups, _lib.ptr(args[2]), _lib.ptr(args[3]), _lib.ptr(r.cuda()) if res else None, _lib.ptr(out), _lib.current_stream())
    d = (out.cpu().double() - want).abs()
    print("%-6s %-40s max %.2e rms %.2e (|want| max %.2f)" % ("legacy" if os.environ.get("IPDM_CONV_LEGACY") else "ws", (B, C1, C2, H, W, Cout, act, res), d.max(), (d**2).mean().sqrt(), want.abs().max()))
for c in [(1, 64, 0, 64, 64, 64, 0, False), (1, 64, 0, 64, 64, 64, 2, True), (1, 128, 64, 32, 32, 64, 2, False), (1, 128, 0, 40, 24, 128, 2, True), (1, 20, 0, 40, 24, 64, 2, False), (2, 256, 0, 16, 16, 256, 2, True)]:
    run(*c)
