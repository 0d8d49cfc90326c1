"""Which evaluation is closer to the exact result?  Compares the HIP outputs dumped by tools/save_unet_out.py (default
exact-f32 mode and the opt-in split-bf16 mode) and the torch-CPU float32 oracle against a float64 evaluation of the same
network (CPU, build container):  python tools/unet_fp64_study.py [img|proj]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import synth
from oracle import unet as ou
torch.set_num_threads(8)
FULL = {"img": (dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(8, 16), channel_mult=(1, 1, 2, 2, 4, 4)), (1, 1, 512, 512)),
        "proj": (dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(16, 32), channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4)), (1, 1, 2000, 912))}
for which in (sys.argv[1:] or ["img"]):
    kw, shape = FULL[which]
    cfg = ou.UNetConfig(**kw)
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(ou.param_shapes(cfg), seed=6).items()}
    x = torch.from_numpy(synth.hash_normal(shape, 401))
    o64 = ou.unet_forward(cfg, {k: v.double() for k, v in sd.items()}, x.double(), 13).numpy()
    o32 = ou.unet_forward(cfg, sd, x, 13).numpy()
    scale = np.abs(o64).max()
    rows = [("torch-CPU float32 oracle", o32)]
    for tag in ("f32", "x6"):
        p = "gpurun_out/unet_%s_%s.npy" % (which, tag)
        if os.path.isfile(p):
            rows.append(("HIP " + tag, np.load(p)))
    print("%s UNet forward, |out|max = %.3f; error against the float64 evaluation:" % (which, scale))
    for name, o in rows:
        d = np.abs(o.astype(np.float64) - o64)
        print("  %-26s max %.3e  mean %.3e  rms %.3e" % (name, d.max(), d.mean(), np.sqrt((d ** 2).mean())))
