"""Times whole UNet forwards of the two production configurations with synthetic weights (GPU box):
   python tools/time_forward.py [B] [iters] [option=value ...]
Every `option=value` (a per-call library switch, e.g. conv_no_wino=1) is timed beside the default, interleaved rounds in ONE
process (best of the rounds); the outputs of all arms must be bit-equal when the switch does not change the arithmetic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ipdm_pytorch_amd import _lib, synth
from ipdm_pytorch_amd.unet import UNetModel
from oracle import unet as ou

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
arms = [("default", None, 0)] + [(a, a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[3:]]
for name, cfg, shape in (("img", ou.UNetConfig(), (B, 1, 512, 512)),
                         ("proj", ou.UNetConfig(attention_resolutions=(16, 32), channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4)), (B, 1, 2000, 912))):
    kw = {k: getattr(cfg, k) for k in ("in_channels", "model_channels", "out_channels", "num_res_blocks", "attention_resolutions",
                                       "channel_mult", "num_heads")}
    net = UNetModel(**kw).to("cuda")
    sd = synth.synth_state_dict(ou.param_shapes(cfg), seed=1)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = torch.from_numpy(synth.hash_normal(shape, 3)).to("cuda")
    best, outs = {}, {}
    for rnd in range(3 if len(arms) > 1 else 1):
        for tag, opt, val in arms:
            if opt:
                _lib.set_option(opt, val)
            y = net(x, 7)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                y = net(x, 7)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
            if opt:
                _lib.set_option(opt, 0)
            best[tag] = min(best.get(tag, 1e9), dt)
            outs[tag] = y.clone()
    for tag, _, _ in arms:
        print("%-5s B=%d  %-22s %.3f ms / forward   checksum %.6f   equals default: %s" % (
            name, B, tag, best[tag] * 1e3, float(outs[tag].double().abs().mean()), bool(torch.equal(outs[tag], outs["default"]))), flush=True)
