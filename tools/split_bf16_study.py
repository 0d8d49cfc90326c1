"""Numerical study (CPU, oracle only): what a 3-term or 6-term split-bf16 evaluation of every convolution would do to
the end-to-end result of the smoke pipeline (reduced UNets, real FBP geometry) -- data for the decision recorded in
DESIGN.md; nothing here is on the product path.   python tools/split_bf16_study.py"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
import ipdm_pytorch_amd
from ipdm_pytorch_amd import synth
from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
from oracle import pipeline as op, diffusion as od, unet as ou

torch.set_num_threads(8)
opt = default_cfg([])
cfg_load(mayo_test_options(), opt.__dict__)
cfg_load(dict(device="cpu", t_start_proj=[2, 2], t_start_img=[2], ultra_img_denoise=True), opt.__dict__)
sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(1)), seed=1)
shape_p, shape_i = (1, 1, 2000, 912), (1, 1, 512, 512)
# same number / order of draws as the pipeline makes (q_sample + steps per pass; proj 2 passes, img 1 + ultra 3)
draws = [torch.from_numpy(synth.hash_normal(shape_p, 7000 + k)) for k in range(6)] + \
        [torch.from_numpy(synth.hash_normal(shape_i, 8000 + k)) for k in range(3 + 18)]
inputs = dict(opt=copy.deepcopy(opt.__dict__), ldproj=sino, noise=draws)
truth = od.miu2pixel(torch.from_numpy(synth.rasterize(synth.ellipse_phantom(1)))).numpy()
orig_conv = F.conv2d


def split(x, terms):
    parts, r = [], x
    for _ in range(terms):
        p = r.bfloat16().float()
        parts.append(p)
        r = r - p
    return parts


def make_conv(nx):
    def conv(x, w, b=None, stride=1, padding=0, *a, **k):
        xs, ws = split(x, nx), split(w, nx)
        y = None
        for i in range(nx):
            for j in range(nx):
                if i + j <= nx - 1:            # 3 terms for nx=2, 6 terms for nx=3
                    t = orig_conv(xs[i], ws[j], None, stride, padding)
                    y = t if y is None else y + t
        return y if b is None else y + b.view(1, -1, 1, 1)
    return conv


res = {}
for name, fn in (("f32", orig_conv), ("bf16x3 (2-way split, 3 terms)", make_conv(2)), ("bf16x6 (3-way split, 6 terms)", make_conv(3))):
    F.conv2d = fn
    out = op.smoke_pipeline_oracle(dict(inputs, noise=list(draws)))
    res[name] = out
    p = od.psnr(truth, od.miu2pixel(torch.from_numpy(out[0, 0])).numpy())
    d = np.abs(out - res["f32"])
    print("%-34s PSNR %.6f dB   rel dPSNR %.2e   max|d| %.2e  rms %.2e" % (
        name, p, abs(p - od.psnr(truth, od.miu2pixel(torch.from_numpy(res['f32'][0, 0])).numpy())) / p, d.max(), np.sqrt((d ** 2).mean())))
F.conv2d = orig_conv
