"""Does a forward read workspace bytes it has not written?  The UNet workspace (torch.empty: uninitialised) is filled with a pattern
before every forward -- zeros, NaNs, 1e30, small numbers -- and the headline batch of two sampled under each.
usage: dbg_bf16x3_ws.py <option 0|1> [short]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ipdm_pytorch_amd import _lib, synth, unet
from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser
DEV = "cuda:0"
on = int(sys.argv[1])
short = len(sys.argv) > 2
over = dict(t_start_proj=[2, 2], t_start_img=[2], ultra_img_denoise=True) if short else dict(t_start_proj=[15, 15, 15], t_start_img=[15], ultra_img_denoise=True)
opt = default_cfg([])
cfg_load(mayo_test_options(), opt.__dict__)
cfg_load(dict(over, device=DEV), opt.__dict__)
sinos = np.stack([synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(p)), seed=p) for p in (0, 1)])
FILL = [None]
plain_ws = unet.UNetModel.workspace


def poisoned(self, B, H, W):
    ws = plain_ws(self, B, H, W)
    if FILL[0] is not None:
        ws.view(torch.float32)[: ws.numel() // 4].fill_(FILL[0])
    return ws


unet.UNetModel.workspace = poisoned


def run(lo, hi):
    den = progressive_domain_denoiser(opt, seed=1234, slice_id0=lo)
    den.data_sample_load(ldproj=torch.from_numpy(sinos[lo:hi])[:, None])
    out = den.progressive_denoiser(sharpen_num=70).cpu().numpy()
    del den
    return out


_lib.set_option("conv_bf16x3", on)
FILL[0] = 0.0
ref = run(0, 2)
for v in (float("nan"), 1e30, 1e-3, 1.0, None, 0.0):
    FILL[0] = v
    got = run(0, 2)
    print("option %d workspace filled with %-6s before every forward: against the zero-filled run %s, %d pixels differ, %d not finite" % (
        on, v, "%.2e" % float(np.nanmax(np.abs(got - ref))), int((got != ref).sum()), int((~np.isfinite(got)).sum())), flush=True)
