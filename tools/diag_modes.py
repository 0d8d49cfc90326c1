#!/usr/bin/env python
"""One forward of both production UNets (full size, B = 1) under the per-call kernel switches, against the default path:
max-abs / rms differences.  Every switch selects another evaluation of the SAME function, so the differences must stay at
float32 rounding level (1e-6 .. 1e-5 of the output scale).   python tools/diag_modes.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
from ipdm_pytorch_amd import _lib, synth       # noqa: E402
from ipdm_pytorch_amd.unet import UNetModel    # noqa: E402

DEV = "cuda:0"
CFGS = {"img": (dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(8, 16), channel_mult=(1, 1, 2, 2, 4, 4)), (1, 1, 512, 512)),
        "proj": (dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(16, 32),
                      channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4)), (1, 1, 2000, 912))}
for name, (kw, shape) in CFGS.items():
    net = UNetModel(**kw).to(DEV)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(net._shapes, seed=5).items()})
    x = torch.from_numpy(synth.hash_normal(shape, 400)).to(DEV)
    base = net(x, 11).cpu()
    sc = float(base.abs().max())
    for opt, val in (("conv_no_up2", 1), ("conv_no_wino", 1), ("direct_no_skip_fuse", 1), ("unet_transpose", 0), ("unet_transpose", 1),
                     ("gn_unfused", 1), ("conv1x1_no_quarter", 1), ("conv_nm", 2)):
        with _lib.option(opt, val):
            o = net(x, 11).cpu()
        e = (o - base).abs()
        print("%-5s %-22s = %d : max %.2e rms %.2e (scale %.2f, rel-max %.1e)" % (name, opt, val, float(e.max()), float((e ** 2).mean().sqrt()), sc, float(e.max()) / sc), flush=True)
