"""Option conv_bf16x3: a conv_wino3 launch while the HOST performs a first use behind it -- the first launch of a kernel from a
translation unit of this library that has not been loaded yet (its code object is uploaded while conv_wino3 runs), or of a torch
operator.  Every launch against a quiet reference, bit for bit.  usage: dbg_bf16x3_lazy.py <option 0|1>"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ipdm_pytorch_amd import _lib, synth
from oracle import unet as ou
DEV = "cuda:0"
on = int(sys.argv[1])
_lib.set_option("conv_bf16x3", on)
code = _lib.lib().ipdm_conv_kernel_code


class Conv:
    def __init__(self, B, C1, C2, Hs, Ws, H, W, Cout, ks, stride, act, res, seed):
        self.g = (B, C1, C2, Hs, Ws, H, W, Cout, ks, stride, act)
        Cin = C1 + C2
        Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
        self.x1 = torch.from_numpy(synth.hash_normal((B, C1, Hs, Ws), seed)).to(DEV)
        self.x2 = torch.from_numpy(synth.hash_normal((B, C2, Hs, Ws), seed + 1)).to(DEV) if C2 else None
        self.r = torch.from_numpy(synth.hash_normal((B, Cout, Ho, Wo), seed + 6)).to(DEV) if res else None
        self.w, self.b, self.ga, self.be = (np.ascontiguousarray(t, dtype=np.float32) for t in (
            synth.hash_normal((Cout, Cin, ks, ks), seed + 2) / np.sqrt(Cin * ks * ks), synth.hash_normal((Cout,), seed + 3),
            synth.hash_uniform((Cin,), seed + 4) + 0.5, synth.hash_normal((Cin,), seed + 5) * 0.2))
        self.shape = (B, Cout, Ho, Wo)
        self.code = code(B, Cout, Cin, ks, stride, H, W)

    def __call__(self):
        B, C1, C2, Hs, Ws, H, W, Cout, ks, stride, act = self.g
        out = torch.full(self.shape, float("nan"), device=DEV)
        _lib.call("ipdm_op_conv2d", _lib.ptr(self.x1), C1, _lib.ptr(self.x2), C2, B, Hs, Ws, H, W, _lib.ptr(self.w), _lib.ptr(self.b), Cout, ks, stride,
                  act, ou.gn_groups(C1 + C2), _lib.ptr(self.ga), _lib.ptr(self.be), _lib.ptr(self.r), _lib.ptr(out), _lib.current_stream())
        return out


main = Conv(4, 256, 0, 256, 256, 256, 256, 256, 3, 1, 2, True, 11)        # ~2 ms of conv_wino3 (the instantiation that adds a residual)
assert main.code == (12 if on else 2), main.code
z = torch.randn(1 << 20, device=DEV)
m = torch.randn(512, 512, device=DEV)
pre = [("ipdm 1x1 (conv_pw)", Conv(1, 256, 0, 64, 64, 64, 64, 768, 1, 1, 1, False, 21)),
       ("ipdm 3x3 stride 2 (conv_ws)", Conv(1, 128, 0, 64, 64, 64, 64, 128, 3, 2, 0, False, 22)),
       ("ipdm 8-channel 3x3 (conv_direct)", Conv(1, 8, 0, 64, 64, 64, 64, 8, 3, 1, 2, False, 23)),
       ("ipdm upsample 3x3 (conv_wup2)", Conv(1, 128, 0, 32, 32, 64, 64, 128, 3, 1, 0, False, 24)),
       ("ipdm 16 -> 128 3x3 (conv_ws)", Conv(1, 16, 0, 64, 64, 64, 64, 128, 3, 1, 2, False, 25))]
# (the triggers' tensors are uploaded HERE -- an upload waits for the stream; behind conv_wino3 only the launch itself happens: option lookup,
#  weight packing on the host, the code object's upload, the launch)
triggers = [(n + " code %d" % c.code, c) for n, c in pre] + [
            ("torch sort", lambda: z.sort()), ("torch cumsum", lambda: z.cumsum(0)), ("torch fft", lambda: torch.fft.rfft(z)),
            ("torch matmul", lambda: m @ m), ("torch erfinv", lambda: z.erfinv()), ("torch topk", lambda: z.topk(5)),
            ("torch conv2d", lambda: torch.nn.functional.conv2d(m[None, None], m[None, None, :3, :3]))]
# (the Conv objects of the triggers are built INSIDE the lambda: uploads and all; build the tensors first, launch behind conv_wino3)
ref = main()
torch.cuda.synchronize()
assert torch.equal(main(), ref)
torch.cuda.synchronize()
bad = 0
for name, trig in triggers:
    got = main()
    trig()
    torch.cuda.synchronize()
    d = (got - ref).abs().max().item()
    again = main()
    torch.cuda.synchronize()
    d2 = (again - ref).abs().max().item()
    bad += int(d != 0) + int(d2 != 0)
    print("option %d: conv_wino3 with a first use behind it (%s): %.2e; the launch after it: %.2e" % (on, name, d, d2), flush=True)
print("option %d: %d launches differ" % (on, bad))
