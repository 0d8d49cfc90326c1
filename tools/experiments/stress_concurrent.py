"""A Winograd convolution while OTHER kernels share the chip (a second stream: element-wise passes over a large tensor, short
workgroups that take SIMD slots and memory bandwidth between the persistent workgroups' waves): every launch against the result of
a quiet launch, bit for bit.  usage: stress_concurrent.py <option conv_bf16x3 0|1> [launches]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ipdm_pytorch_amd import _lib, synth
from oracle import unet as ou
DEV = "cuda:0"
on = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
_lib.set_option("conv_bf16x3", on)
code = _lib.lib().ipdm_conv_kernel_code
side = torch.cuda.Stream()
main = torch.cuda.Stream()
z = torch.ones(48 << 20, device=DEV)
zs = [torch.ones(1 << 14, device=DEV) for _ in range(8)]
total_bad = 0
for (B, C1, C2, H, W, Cout, act, res) in [(2, 128, 0, 250, 114, 128, 2, True), (2, 128, 16, 250, 114, 128, 2, False), (2, 128, 128, 125, 57, 128, 2, True),
                                          (2, 256, 0, 64, 64, 256, 2, True), (8, 128, 0, 128, 128, 128, 2, True), (1, 128, 0, 130, 250, 128, 2, True)]:
    Cin, seed = C1 + C2, 77 + C1 + C2 + H
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed)).to(DEV)
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)).to(DEV) if C2 else None
    rd = torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)).to(DEV) if res else None
    wn, bn, gn_, ben = (np.ascontiguousarray(t, dtype=np.float32) for t in (
        synth.hash_normal((Cout, Cin, 3, 3), seed + 2) / np.sqrt(Cin * 9), synth.hash_normal((Cout,), seed + 3),
        synth.hash_uniform((Cin,), seed + 4) + 0.5, synth.hash_normal((Cin,), seed + 5) * 0.2))

    def once():
        out = torch.full((B, Cout, H, W), float("nan"), device=DEV)
        _lib.call("ipdm_op_conv2d", _lib.ptr(x1), C1, _lib.ptr(x2), C2, B, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1,
                  act, ou.gn_groups(Cin), _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
        return out
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        ref = once()
        quiet_bad = sum(int(not torch.equal(once(), ref)) for _ in range(5))
    torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for mode in ("large", "small"):
        for k in range(reps):
            with torch.cuda.stream(side):
                if mode == "large":
                    for _ in range(3):
                        z.mul_(1.0)
                else:
                    for _ in range(40):
                        zs[k % 8].mul_(1.0)
            with torch.cuda.stream(main):
                got = once()
            torch.cuda.synchronize()
            if not torch.equal(got, ref):
                bad += 1
                worst = max(worst, (got - ref).abs().max().item())
    total_bad += bad
    print("option %d code %2d %s: quiet repeats differing %d/5; under a busy second stream %d/%d differ (worst %.2e)" % (
        on, code(B, Cout, Cin, 3, 1, H, W), (B, C1, C2, H, W, Cout, act, res), quiet_bad, bad, 2 * reps, worst), flush=True)
print("option %d: %d launches differed" % (on, total_bad))
