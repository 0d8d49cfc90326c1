#!/usr/bin/env python
"""Where does the end-to-end distance to the float64 arbiter come from?  The reduced smoke pipeline with every stored
iterate kept, HIP vs oracle-f32 vs oracle-f64 stage by stage (proj passes, FBP of the last one, img passes).
  python tools/arbiter_stages.py"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np        # noqa: E402
import torch              # noqa: E402
from ipdm_pytorch_amd import synth    # noqa: E402
from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options    # noqa: E402
from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, SMOKE_PROJ, SMOKE_IMG, _RecordingNoise    # noqa: E402
from ipdm_pytorch_amd.diffusion import NoiseSource    # noqa: E402
from ipdm_pytorch_amd.unet import UNetModel    # noqa: E402
from oracle import pipeline as op, unet as ou    # noqa: E402

DEV = "cuda:0"


def dist(a, b):
    e = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    return float(e.max()), float(np.sqrt((e ** 2).mean()))


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(device=DEV, t_start_proj=[2, 2], t_start_img=[2], ultra_img_denoise=True, save_it_state_proj=True,
                  save_it_state_img=True), opt.__dict__)
    den = progressive_domain_denoiser(opt, seed=11)
    den.proj_model = UNetModel(**SMOKE_PROJ).to(DEV)
    den.img_model = UNetModel(**SMOKE_IMG).to(DEV)
    sd_p = synth.synth_state_dict(den.proj_model._shapes, seed=21)
    sd_i = synth.synth_state_dict(den.img_model._shapes, seed=22)
    den.proj_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_p.items()})
    den.img_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_i.items()})
    sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(1)), seed=1)
    den.data_sample_load(ldproj=torch.from_numpy(sino)[None, None])
    rec = _RecordingNoise(NoiseSource(11, 0))
    den.noise = rec
    den.progressive_denoiser(save_proj_state=True, sharpen_num=70)
    draws = [z.cpu() for z in rec.draws]
    cfg_p = ou.UNetConfig(1, 16, 1, attention_resolutions=(16,), channel_mult=(0.25, 0.25, 0.5, 1, 2, 4), num_heads=1)
    cfg_i = ou.UNetConfig(1, 16, 1, attention_resolutions=(8,), channel_mult=(1, 1, 2, 2, 4), num_heads=1)
    mids = {}
    for dt in (torch.float32, torch.float64):
        it = iter(draws)
        _, mid = op.progressive_slice(dict(opt.__dict__), cfg_p, {k: torch.from_numpy(v).to(dt) for k, v in sd_p.items()}, cfg_i,
                                      {k: torch.from_numpy(v).to(dt) for k, v in sd_i.items()},
                                      torch.from_numpy(sino)[None, None].to(dt), lambda: next(it).to(dt), sharpen_num=70)
        mids[dt] = mid
    m32, m64 = mids[torch.float32], mids[torch.float64]

    def line(name, hip, a32, a64):
        h, c, f = dist(hip, a64), dist(a32, a64), dist(hip, a32)
        print("%-22s |hip-f64| max %.2e rms %.2e | |cpu32-f64| max %.2e rms %.2e | |hip-cpu32| max %.2e rms %.2e | ratio rms %.2f scale %.2f" % (
            name, h[0], h[1], c[0], c[1], f[0], f[1], h[1] / max(c[1], 1e-30), float(np.abs(np.asarray(a64)).max())), flush=True)
    for k in range(len(m32["proj"])):
        line("proj iter_%d" % (k + 1), den.proj_denoise_result[k + 1], m32["proj"][k].numpy(), m64["proj"][k].numpy())
    n = len(den.proj_denoise_convert2img_result)
    line("fbp(last proj)", den.proj_denoise_convert2img_result[n], m32["fbp"].numpy(), m64["fbp"].numpy())
    for k in range(len(m32["img"])):
        line("img iter_%d" % (k + 1), den.progressive_denoise_result[k + 1], m32["img"][k].numpy(), m64["img"][k].numpy())
    # single forwards of the two reduced networks on the same input
    for tag, kw, shape in (("unet smoke-proj", SMOKE_PROJ, (1, 1, 2000, 912)), ("unet smoke-img", SMOKE_IMG, (1, 1, 512, 512))):
        net = den.proj_model if "proj" in tag else den.img_model
        sd = sd_p if "proj" in tag else sd_i
        cfg = cfg_p if "proj" in tag else cfg_i
        x = torch.from_numpy(synth.hash_normal(shape, 401))
        got = net(x.to(DEV), 1).cpu().numpy()
        w32 = ou.unet_forward(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, x, 1).numpy()
        w64 = ou.unet_forward(cfg, {k: torch.from_numpy(v).double() for k, v in sd.items()}, x.double(), 1).numpy()
        line(tag, got, w32, w64)


if __name__ == "__main__":
    main()
