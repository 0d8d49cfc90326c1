import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
import ipdm_pytorch_amd
from ipdm_pytorch_amd import _lib, synth
B, Cin, H, W, Cout = 1, 8, 32, 32, 64
x = torch.from_numpy(synth.hash_normal((B, Cin, H, W), 1))
w = torch.from_numpy(synth.hash_normal((Cout, Cin, 3, 3), 2)) * 0.1
b = torch.from_numpy(synth.hash_normal((Cout,), 3))
want = F.conv2d(x, w, b, padding=1)
out = torch.full(tuple(want.shape), -77.0, device="cuda")
xd = x.cuda()
wn, bn = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(b.numpy())
_lib.call("ipdm_op_conv2d", _lib.ptr(xd), Cin, None, 0, B, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1, 0, 1, None, None, None, _lib.ptr(out), _lib.current_stream())
got = out.cpu()
err = (got - want).abs()
print("max err", err.max().item(), "untouched", (got == -77).sum().item(), "of", got.numel())
bad = (err > 1e-3)
print("bad per cout:", bad.sum(dim=(0, 2, 3)).tolist())
print("bad per row:", bad.sum(dim=(0, 1, 3)).tolist())
print("bad per col:", bad.sum(dim=(0, 1, 2)).tolist())
wb = F.conv2d(x, w, None, padding=1)
print("err vs no-bias:", (got - wb).abs().max().item())
print("got[0,0,0,:6]", got[0, 0, 0, :6].tolist(), "want", want[0, 0, 0, :6].tolist(), "nobias", wb[0,0,0,:6].tolist())
for c in (1, 2, 3, 5, 8, 9):
    d = [(got[0, c] - want[0, c2]).abs().max().item() for c2 in range(Cout)]
    best = int(np.argmin(d))
    print("got cout", c, "matches want cout", best, "err", d[best])
