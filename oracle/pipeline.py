"""CPU oracle: restatement of the harness orchestration proj_denoiser -> FBP -> tensor_sharpen ->
img_denoiser (+ultra) (Utils/train_test_utils.py:421-567) with per-slice semantics.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product path."""
import numpy as np
import torch

from . import diffusion as od
from . import fbp as of
from . import unet as ou


def progressive_slice(opt, cfg_p, sd_p, cfg_i, sd_i, ldproj, noise_fn, geo=None, sharpen_num=42):
    """progressive_denoiser for ONE slice: ldproj [1,1,n_views,n_det] float32 tensor -> [1,1,G,G].
    `opt` is a dict with the reference's option names.  Returns (final, dict of intermediates)."""
    geo = geo or of.FBPGeometry()
    sch_p = od.Schedule(opt["timesteps_proj"], opt["schedule_power_proj"])
    sch_i = od.Schedule(opt["timesteps_img"], opt["schedule_power_img"])
    res_p, ns = od.guided_reverse_process_slice(
        sch_p, lambda x, t: ou.unet_forward(cfg_p, sd_p, x, t), ldproj, t_start=opt["t_start_proj"], clip=opt["clip_proj"],
        lambda_ratio=opt["lambda_ratio_proj"], eta=opt["eta_proj"], mode="proj",
        constant_guidance=opt["constant_guidance_proj"], noise_fn=noise_fn, kernel_size=opt["kernel_size_proj"],
        amplitude=opt["amplitude_proj"])
    G = 10 if opt["clip_proj"] else 1
    fbp = of.convert64 if ldproj.dtype == torch.float64 else of.convert          # float64 input: arbiter run (below)
    fbp_img = torch.from_numpy(fbp(geo, (G * res_p[-1][:, 0]).numpy()))[:, None]
    x = od.tensor_sharpen(fbp_img, sharpen_num if (opt["convertor"] == "FBP" and opt["fbp_sharpen"]) else -1)
    eps_i = lambda xx, t: ou.unet_forward(cfg_i, sd_i, xx, t)   # noqa: E731
    res_i, _ = od.guided_reverse_process_slice(
        sch_i, eps_i, x, t_start=opt["t_start_img"], clip=opt["clip_img"], lambda_ratio=opt["lambda_ratio_img"],
        eta=opt["eta_img"], mode="img", constant_guidance=opt["constant_guidance_img"], noise_fn=noise_fn, ldct=x,
        kernel_size=opt["kernel_size_img"], amplitude=opt["amplitude_img"], noise_strength_in=ns)
    if opt["ultra_img_denoise"]:
        res_u, _ = od.guided_reverse_process_slice(
            sch_i, eps_i, res_i[-1], t_start=[5, 5, 5], clip=opt["clip_img"], lambda_ratio=opt["lambda_ratio_img"],
            eta=0.6, mode="img", constant_guidance=0.6, noise_fn=noise_fn, ldct=x, kernel_size=opt["kernel_size_img"],
            amplitude=opt["amplitude_img"], noise_strength_in=ns)
        res_i = res_i + res_u
    return res_i[-1], dict(proj=res_p, fbp=fbp_img, sharpened=x, img=res_i)


def smoke_pipeline_oracle(inputs, dtype=torch.float32):
    """Replays ipdm_pytorch_amd.denoiser.smoke_pipeline on the CPU with the recorded noise.

    dtype=torch.float64 is the ARBITER run: the same function -- the same float32 inputs, weights, noise draws, schedule
    constants and guidance maps -- evaluated in double precision (UNet, statistics, FBP, sharpen), i.e. to ~1e-13 the value
    that both float32 evaluations (this oracle's and the HIP library's) approximate.  Random-weight networks amplify
    float32 rounding ~100x end to end, so |HIP - oracle32| alone cannot say which side is off; the distances of each to
    the arbiter can."""
    from ipdm_pytorch_amd import synth
    opt = inputs["opt"]
    cfg_p = ou.UNetConfig(1, 16, 1, attention_resolutions=(16,), channel_mult=(0.25, 0.25, 0.5, 1, 2, 4), num_heads=1)
    cfg_i = ou.UNetConfig(1, 16, 1, attention_resolutions=(8,), channel_mult=(1, 1, 2, 2, 4), num_heads=1)
    sd_p = {k: torch.from_numpy(v).to(dtype) for k, v in synth.synth_state_dict(ou.param_shapes(cfg_p), seed=21).items()}
    sd_i = {k: torch.from_numpy(v).to(dtype) for k, v in synth.synth_state_dict(ou.param_shapes(cfg_i), seed=22).items()}
    draws = iter(inputs["noise"])
    out, _ = progressive_slice(opt, cfg_p, sd_p, cfg_i, sd_i, torch.from_numpy(inputs["ldproj"])[None, None].to(dtype),
                               lambda: torch.as_tensor(next(draws)).to(dtype), sharpen_num=70)
    return out.numpy()
