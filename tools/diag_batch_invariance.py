"""Which switch breaks `slice of a batch == slice alone` at full size?  (dev aid, GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import synth
from ipdm_pytorch_amd.unet import UNetModel

DEV = "cuda:0"
NETS = {"img": (dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(8, 16), channel_mult=(1, 1, 2, 2, 4, 4)), (512, 512)),
        "proj": (dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(16, 32),
                      channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4)), (2000, 912))}
for name, (kw, (H, W)) in NETS.items():
    net = UNetModel(**kw).to(DEV)
    x = torch.from_numpy(synth.hash_normal((8, 1, H, W), 5)).to(DEV)
    a = net(x, 7)
    res = []
    for B in (1, 2, 4):
        b = net(x[:B].contiguous(), 7)
        res.append("B=%d %s (%.1e)" % (B, bool(torch.equal(a[:B], b)), float((a[:B] - b).abs().max())))
    print(name, {k: v for k, v in os.environ.items() if k.startswith("IPDM_")}, " | ".join(res), flush=True)
