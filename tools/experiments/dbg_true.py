import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_parity as T
from ipdm_pytorch_amd import _lib
ou, synth, DEV = T.ou, T.synth, T.DEV
kw, shape = T.FULL_PROJ, (1, 1, 2000, 912)
net, sd = T._native_unet(kw, 6)
x = torch.from_numpy(synth.hash_normal(shape, 401))
torch.set_num_threads(32)
want = ou.unet_forward(ou.UNetConfig(**kw), sd, x, 13)
want64 = ou.unet_forward(ou.UNetConfig(**kw), {k: v.double() for k, v in sd.items()}, x.double(), 13) if hasattr(ou, "unet_forward") else None
for name, opts in (("default", {}), ("conv_no_wup2", {"conv_no_wup2": 1}), ("conv_no_up2", {"conv_no_up2": 1})):
    import contextlib
    with contextlib.ExitStack() as st:
        for k, v in opts.items(): st.enter_context(_lib.option(k, v))
        got = net(x.to(DEV), 13).cpu()
    e = (got - want).abs()
    e64 = (got.double() - want64).abs()
    print(name, "max err vs oracle f32 %.3e  rms %.3e | vs oracle f64 max %.3e rms %.3e | max|want| %.3f" % (e.max(), e.pow(2).mean().sqrt(), e64.max(), e64.pow(2).mean().sqrt(), want.abs().max()), flush=True)
e = (want.double() - want64).abs()
print("oracle f32 vs f64: max %.3e rms %.3e" % (e.max(), e.pow(2).mean().sqrt()))
