"""Case definitions shared by tests/golden/make_golden.py (reference side) and the tests (oracle /
HIP side).  Data only."""
import torch

from ipdm_pytorch_amd import synth

SMALL_CFGS = {
    "a": dict(in_channels=1, model_channels=16, out_channels=1, num_res_blocks=2, attention_resolutions=(2, 4),
              channel_mult=(1, 1, 2, 4), num_heads=1),
    "b": dict(in_channels=1, model_channels=48, out_channels=1, num_res_blocks=1, attention_resolutions=(4,),
              channel_mult=(0.25, 0.5, 0.75, 4 / 3), num_heads=1),
    "c": dict(in_channels=1, model_channels=8, out_channels=1, num_res_blocks=1, attention_resolutions=(1, 2),
              channel_mult=(1, 2, 4), num_heads=4),
    # wide levels at a small size (32 / 64 / 128 channels at 16x24, 8x12, 4x6): both Upsample convolutions double their
    # input exactly and run on the MFMA kernels, i.e. in the parity form with parity-planar outputs
    "d": dict(in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=(4,),
              channel_mult=(1, 2, 4), num_heads=2),
}
SMALL_SHAPES = {"a": (2, 1, 24, 20), "b": (1, 1, 23, 19), "c": (1, 1, 12, 10), "d": (2, 1, 16, 24)}

LOOP_CFG = dict(in_channels=1, model_channels=16, out_channels=1, num_res_blocks=1, attention_resolutions=(4,),
                channel_mult=(1, 1, 2, 4), num_heads=1)
LOOP_CASES = {
    "img_const": ("img", (1, 1, 32, 32), 1, dict(t_start=[3, 2], clip=True, lambda_ratio=10, eta=0.7,
                                                 constant_guidance=0.45)),
    "img_adapt": ("img", (1, 1, 32, 32), 1, dict(t_start=[3, 3], clip=True, lambda_ratio=10, eta=0.7,
                                                 constant_guidance=None)),
    "proj_adapt": ("proj", (1, 1, 40, 24), 5, dict(t_start=[3, 3, 2], clip=False, lambda_ratio=1, eta=0.5,
                                                   constant_guidance=None)),
    "proj_const": ("proj", (1, 1, 40, 24), 5, dict(t_start=[2, 2], clip=True, lambda_ratio=1, eta=0.5,
                                                   constant_guidance=0.3)),
}

# sparse (DDIM) sampler: (shape, schedule power, kwargs of sparse_guided_reverse_process)
SPARSE_CASES = {
    "img": ((1, 1, 32, 32), 1, dict(t_start=[4, 3, 3], ddim_timesteps=[1, 2, 2], condition_lambda_max=0.5,
                                    condition_lambda_min=0.3, eta=0.7, clip_denoised=True, ddim_eta=0.0)),
    "proj": ((1, 1, 40, 24), 5, dict(t_start=[5, 4], ddim_timesteps=[2, 2], condition_lambda_max=0.49,
                                     condition_lambda_min=0.35, eta=0.5, clip_denoised=False, ddim_eta=0.0)),
    "img_eta": ((1, 1, 32, 32), 1, dict(t_start=[6, 4], ddim_timesteps=[3, 2], condition_lambda_max=0.5,
                                        condition_lambda_min=0.3, eta=0.6, clip_denoised=True, ddim_eta=0.3)),
}

# adaptive pass schedule (t_start=None, Model/model.py:532-536,582-613,639-640): (mode, shape, schedule power,
# amplitude, noise_strength handed in, kwargs).  proj picks its branch from delt.max() (amplitude chosen well inside
# each branch: low <4.5, mid [4.5,30), high >=30) and reports it; img takes the branch from `noise_strength`.
ADAPT_CASES = {
    "proj_low": ("proj", (1, 1, 40, 24), 5, 7, None, dict(clip=False, lambda_ratio=1, eta=0.5)),
    "proj_mid": ("proj", (1, 1, 40, 24), 5, 20, None, dict(clip=False, lambda_ratio=1, eta=0.5)),
    "proj_high": ("proj", (1, 1, 40, 24), 5, 45, None, dict(clip=True, lambda_ratio=1, eta=0.5)),
    "img_high": ("img", (1, 1, 32, 32), 1, 30, "high", dict(clip=True, lambda_ratio=10, eta=0.7)),
    "img_mid": ("img", (1, 1, 32, 32), 1, 30, "mid", dict(clip=True, lambda_ratio=10, eta=0.7)),
    "img_none": ("img", (1, 1, 32, 32), 1, 30, None, dict(clip=False, lambda_ratio=10, eta=0.7)),
}


class noise_feed:
    """Hashed N(0,1) draws in call order: draw k of feed `seed` = hash_normal(shape, seed*1000+k)."""

    def __init__(self, seed, shape):
        self.seed, self.shape, self.count = seed, tuple(shape), 0

    def __call__(self):
        z = torch.from_numpy(synth.hash_normal(self.shape, self.seed * 1000 + self.count))
        self.count += 1
        return z


# End-to-end fixture (tests/golden/pipeline.npz): the reference's own progressive_domain_denoiser driven through
# proj_denoiser -> FBP convertor -> tensor_sharpen -> img_denoiser (+ultra) (Utils/train_test_utils.py:421-567) at the
# TRUE geometry (2000x912 -> 512x512) with reduced UNets.  The harness always builds UNetModel with its default 4 heads
# (Utils/train_test_utils.py:213-245), so the attention levels carry 128 channels (head dim 32).
PIPE_OPT = dict(
    mode="test_prog", convertor="FBP", fbp_sharpen=True, ultra_img_denoise=True, normal=False, benchmark_test=False,
    resume_epochs_proj=0, resume_epochs_img=0,
    model_channels_proj=32, channel_mult_proj=[0.125, 0.125, 0.125, 0.25, 0.5, 4], attention_resolutions_proj=[16],
    model_channels_img=32, channel_mult_img=[1, 1, 1, 2, 4], attention_resolutions_img=[8],
    t_start_proj=[2, 2], t_start_img=[2], sample_method_proj="dense", sample_method_img="dense",
    constant_guidance_proj=None, constant_guidance_img=0.45, save_it_state_proj=False, save_it_state_img=False)
PIPE_SEEDS = dict(proj_weights=71, img_weights=72, phantom=1, dose=1, noise=73)
PIPE_SHARPEN = 70


# BASELINE.json config C1, literally: one 512x512 slice, image domain only, t_start_img=[5], constant guidance 0.45, no ultra pass
# (tests/golden/pipeline_c1.npz from the reference harness's img_denoiser(mode="img_only"), PIPE_OPT networks)
C1_OPT = dict(t_start_img=[5], constant_guidance_img=0.45, ultra_img_denoise=False, save_it_state_img=True)
C1_NOISE_SEED, C1_INPUT_SEED = 76, 75


class hashed_noise:
    """Noise source for the HIP path (next_like) and the oracle (draws(shapes)): draw k = hash_normal(shape of the
    k-th request, seed*1000+k) -- what make_golden's _NoiseFeed hands the reference in place of torch.randn_like."""

    def __init__(self, seed):
        self.seed, self.draw = seed, 0

    def next_like(self, x):
        z = torch.from_numpy(synth.hash_normal(tuple(x.shape), self.seed * 1000 + self.draw)).to(x.device)
        self.draw += 1
        return z


def pipeline_draw_shapes(opt, proj_shape, img_shape):
    """Shapes of the randn draws of one progressive_denoiser() call with fixed t_start lists, in call order."""
    n_proj = sum(t + 1 for t in opt["t_start_proj"])
    n_img = sum(t + 1 for t in opt["t_start_img"]) + (18 if opt["ultra_img_denoise"] else 0)
    return [tuple(proj_shape)] * n_proj + [tuple(img_shape)] * n_img
