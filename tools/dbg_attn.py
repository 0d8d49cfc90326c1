import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import _lib, synth
B, heads, T, d = 1, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 7125, 64
qkv = torch.from_numpy(synth.hash_normal((B, heads * 3 * d, T), 300 + T)) * 1.5
out = torch.empty((B, heads * d, T), device="cuda")
_lib.call("ipdm_op_attention", _lib.ptr(qkv.cuda()), _lib.ptr(out), B, heads, d, T, _lib.current_stream())
q, k, v = qkv.reshape(B * heads, 3 * d, T).chunk(3, dim=1)
scale = 1.0 / np.sqrt(np.sqrt(d))
S = torch.einsum("bct,bcs->bts", (q * scale).double(), (k * scale).double())
attn = S.softmax(dim=-1)
want = torch.einsum("bts,bcs->bct", attn, v.double()).reshape(B, heads * d, T).float()
err = (out.cpu() - want).abs()[0]      # [c, t]
print("max err %.2e  mean %.2e" % (err.max(), err.mean()))
pt = err.max(dim=0).values
idx = torch.argsort(pt, descending=True)[:10]
print("worst queries:", [(int(i), "%.1e" % pt[i], "Smax %.1f" % S[0, i].max(), "pmax %.3f" % attn[0, i].max()) for i in idx])
print("err>1e-5 queries:", int((pt > 1e-5).sum()), "of", T)
