"""Generates the golden vectors in tests/golden/ by IMPORTING THE REFERENCE (read-only, at
/root/reference) through oracle/ref_shim.py and running its own functions on seeded inputs.

Run in the build container only:   python tests/golden/make_golden.py
The fixtures are data (inputs are regenerated from integer hashes by ipdm_pytorch_amd.synth; only
small outputs / index lists are stored).  Nothing here copies reference source.
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
import ipdm_pytorch_amd  # noqa: E402,F401
from ipdm_pytorch_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
M, FB = ref_shim.load()
U = ref_shim.load_curves()
torch.set_num_threads(8)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ 1. schedules
def gen_schedule():
    names = ["sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
             "posterior_log_variance_clipped", "posterior_variance"]
    out = {}
    for p in (1, 5):
        gd = M.GaussianDiffusion(timesteps=1000, beta_schedule="cosine", schedule_power=p)
        t = torch.arange(0, 30)
        tab = np.stack([gd._extract(getattr(gd, n), t, (30, 1)).reshape(-1).numpy() for n in names])
        out["tables_p%d" % p] = tab.astype(np.float32)
    for ts, power in ((15, 1), (15, 10), (5, 10), (20, 1)):
        out["lambda_ts%d_p%d" % (ts, power)] = M.cosine_beta_schedule(ts, schedule_power=power).numpy()
    save("schedule", **out)


# ------------------------------------------------------------------ 2. group rule
def gen_groups():
    chans = np.array([1, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 132, 136, 144, 192, 256, 272, 384, 512, 36, 40, 100])
    groups = np.array([M.norm_layer(int(c)).num_groups for c in chans])
    save("gn_groups", channels=chans, groups=groups)


# ------------------------------------------------------------------ 3. UNet
def ref_unet(cfg, seed):
    net = M.UNetModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed=seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.eval()
    return net, list(shapes.keys())


from tests.golden.cases import SMALL_CFGS, SMALL_SHAPES, LOOP_CFG, LOOP_CASES, SPARSE_CASES  # noqa: E402


def gen_unet():
    out = {}
    for tag, cfg in SMALL_CFGS.items():
        net, keys = ref_unet(cfg, seed=11)
        x = torch.from_numpy(synth.hash_normal(SMALL_SHAPES[tag], 101))
        for t in (0, 7):
            with torch.no_grad():
                y = net(x, torch.full((1,), t, dtype=torch.long))
            out["%s_t%d" % (tag, t)] = y.numpy()
        out[tag + "_keys"] = np.array(keys)
    save("unet_small", **out)


# ------------------------------------------------------------------ 4. single blocks
def gen_ops():
    out = {}
    torch.manual_seed(0)
    # AttentionBlock: C=64, heads=1 (d=64), T = 5*7 = 35 (not a multiple of anything)
    for tag, (C, heads, H, W) in {"attn64": (64, 1, 5, 7), "attn256": (256, 4, 9, 13)}.items():
        blk = M.AttentionBlock(C, num_heads=heads)
        shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
        blk.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=21).items()})
        x = torch.from_numpy(synth.hash_normal((1, C, H, W), 22))
        with torch.no_grad():
            y = blk(x)
        out[tag + "_out"] = y.numpy()
    # Upsample to explicit odd size (nearest index rule) + conv
    up = M.Upsample(8, True)
    shapes = {k: tuple(v.shape) for k, v in up.state_dict().items()}
    up.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=23).items()})
    x = torch.from_numpy(synth.hash_normal((1, 8, 29, 63), 24))
    with torch.no_grad():
        out["up_out"] = up(x, (57, 125)).numpy()
    # nearest index maps for the sizes the proj UNet meets (63->125, 29->57) and a few awkward ones
    for (i, o) in ((63, 125), (29, 57), (125, 250), (57, 114), (3, 7), (5, 12), (500, 2000), (228, 912)):
        src = torch.arange(i, dtype=torch.float32)[None, None, None, :]
        idx = torch.nn.functional.interpolate(src, size=(1, o), mode="nearest").reshape(-1).numpy().astype(np.int32)
        out["nearest_%d_%d" % (i, o)] = idx
    # ResidualBlock with a conv shortcut and awkward channel counts
    rb = M.ResidualBlock(36, 24, 64, 0)
    shapes = {k: tuple(v.shape) for k, v in rb.state_dict().items()}
    rb.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=25).items()})
    x = torch.from_numpy(synth.hash_normal((2, 36, 11, 9), 26))
    emb = torch.from_numpy(synth.hash_normal((1, 64), 27))
    with torch.no_grad():
        out["res_out"] = rb(x, emb).numpy()
    out["res_keys"] = np.array(list(shapes.keys()))
    save("ops", **out)


# ------------------------------------------------------------------ 5. one guided step
class _NoiseFeed:
    """Replaces torch.randn_like inside the reference: hands out hashed normals in call order."""

    def __init__(self, seed):
        self.seed, self.k = seed, 0
        self.log = []

    def __call__(self, x, *a, **k):
        z = torch.from_numpy(synth.hash_normal(tuple(x.shape), self.seed * 1000 + self.k))
        self.k += 1
        return z


def gen_step():
    out = {}
    gd = M.GaussianDiffusion(timesteps=1000, beta_schedule="cosine", schedule_power=5)
    shape = (1, 1, 16, 12)
    x_t = torch.from_numpy(synth.hash_normal(shape, 31)) * 0.3 + 0.5
    x_0 = torch.from_numpy(synth.hash_normal(shape, 32)) * 0.2 + 0.5
    pred = torch.from_numpy(synth.hash_normal(shape, 33)) * 1.7 + 0.1
    lam_small = torch.from_numpy(synth.hash_uniform((1, 1, 4, 3), 34)) * 0.9 + 0.05
    lam_map = torch.nn.functional.interpolate(lam_small, size=shape[-2:], mode="nearest")
    orig = torch.randn_like
    try:
        for tag, (t, lam, clip) in {"scalar_t7": (7, 0.45, True), "scalar_t0": (0, 0.45, True),
                                    "tensor0d_t5": (5, M.cosine_beta_schedule(15, schedule_power=1)[5], False),
                                    "map_t3": (3, lam_map, False)}.items():
            torch.randn_like = _NoiseFeed(35)
            y = gd.p_sample_condition(lambda x, tt: pred, x_t, x_0, torch.full((1,), t, dtype=torch.long),
                                      clip_denoised=clip, lambda_=lam, mode="proj")
            out[tag] = y.numpy()
    finally:
        torch.randn_like = orig
    save("step", **out)


# ------------------------------------------------------------------ 6. guided_reverse_process
def gen_loops():
    out = {}
    net, _ = ref_unet(LOOP_CFG, seed=41)
    curves = {"img": U.curve_init(), "proj": U.proj_curv_init()}
    orig = torch.randn_like
    cases = LOOP_CASES
    try:
        for tag, (mode, shape, power, kw) in cases.items():
            gd = M.GaussianDiffusion(timesteps=1000, beta_schedule="cosine", schedule_power=power)
            if mode == "img":
                img = torch.from_numpy(synth.hash_uniform(shape, 42)) * 0.05 + 0.17     # mu-like values
            else:
                img = torch.from_numpy(synth.hash_uniform(shape, 43)) * 0.6
            ldct = torch.from_numpy(synth.hash_uniform(shape, 44)) * 0.05 + 0.17
            feed = _NoiseFeed(45)
            torch.randn_like = feed
            res, _, ns = gd.guided_reverse_process(
                model=net, img=img, mode=mode, save_states=False, lambda_curve=curves[mode], ldct=ldct,
                kernel_size_img=4, amplitude_img=30, kernel_size_proj=4, amplitude_proj=7, only_convertor=False,
                normal=False, noise_strength=None, transformer=None, **kw)
            out[tag] = np.stack([r.numpy() for r in res])
            out[tag + "_ndraws"] = np.array(feed.k)
    finally:
        torch.randn_like = orig
    save("loops", **out)


# ------------------------------------------------------------------ 6a. adaptive pass schedule (t_start=None)
def gen_adaptive():
    from tests.golden.cases import ADAPT_CASES
    out = {}
    net, _ = ref_unet(LOOP_CFG, seed=41)
    curves = {"img": U.curve_init(), "proj": U.proj_curv_init()}
    orig = torch.randn_like
    try:
        for tag, (mode, shape, power, amp, ns_in, kw) in ADAPT_CASES.items():
            gd = M.GaussianDiffusion(timesteps=1000, beta_schedule="cosine", schedule_power=power)
            if mode == "img":
                img = torch.from_numpy(synth.hash_uniform(shape, 42)) * 0.05 + 0.17
            else:
                img = torch.from_numpy(synth.hash_uniform(shape, 43)) * 0.6
            ldct = torch.from_numpy(synth.hash_uniform(shape, 44)) * 0.05 + 0.17
            feed = _NoiseFeed(48)
            torch.randn_like = feed
            res, _, ns = gd.guided_reverse_process(
                model=net, img=img, mode=mode, t_start=None, save_states=False, lambda_curve=curves[mode], ldct=ldct,
                kernel_size_img=4, amplitude_img=amp, kernel_size_proj=4, amplitude_proj=amp, only_convertor=False,
                normal=False, noise_strength=ns_in, transformer=None, constant_guidance=None, **kw)
            out[tag] = np.stack([r.numpy() for r in res])
            out[tag + "_ndraws"] = np.array(feed.k)
            out[tag + "_ns"] = np.array(str(ns))
            print("  adaptive %-10s draws %3d noise_strength %s" % (tag, feed.k, ns))
    finally:
        torch.randn_like = orig
    save("adaptive", **out)


# ------------------------------------------------------------------ 6b. sparse_guided_reverse_process (DDIM)
def gen_sparse():
    out = {}
    net, _ = ref_unet(LOOP_CFG, seed=41)
    orig = torch.randn_like
    try:
        for tag, (shape, power, kw) in SPARSE_CASES.items():
            gd = M.GaussianDiffusion(timesteps=1000, beta_schedule="cosine", schedule_power=power)
            cond = torch.from_numpy(synth.hash_uniform(shape, 46)) * 0.6
            feed = _NoiseFeed(47)
            torch.randn_like = feed
            res = gd.sparse_guided_reverse_process(model=net, condition=cond, **kw)
            out[tag] = np.stack([r.numpy() for r in res])
            out[tag + "_ndraws"] = np.array(feed.k)
    finally:
        torch.randn_like = orig
    save("sparse", **out)


# ------------------------------------------------------------------ 7. FBP
def fbp_cpu_vectorised(fbp, pj):
    """fbp_cpu (Recon/FBP_kernel.py:166-184) with the pixel loops vectorised in numpy: identical
    float64 per-element arithmetic and the same in-order float32 accumulation over views."""
    BS = pj.shape[0]
    G = fbp.grid.N
    I = np.zeros((BS, G, G), dtype=np.float32)
    for t in range(fbp.M):
        beta = fbp.theta[t] - np.pi / 2
        th = np.pi / 2 + beta + fbp.phi
        alpha = np.arctan(fbp.r * np.sin(th) / (fbp.D + fbp.r * np.cos(th)))
        u = (alpha - fbp.nda[0]) / fbp.da + 0.5
        cur = np.floor(u)
        ok = (0 < cur) & (cur < fbp.N)
        lam = u - cur
        L = fbp.r * np.sin(th) / np.sin(alpha)
        ci = np.where(ok, cur, 1).astype(np.int64)
        for k in range(BS):
            row = pj[k, t]
            inc = ((1 - lam) * row[ci - 1] + lam * row[ci]) / L ** 2
            I[k] = np.where(ok, (I[k] + inc).astype(np.float32), I[k])
    return I


def gen_fbp():
    out = {}
    fbp = FB.FBP("cpu")
    out["theta"] = fbp.theta[::97].copy()
    out["nda"] = fbp.nda[::57].copy()
    out["h_RL"] = fbp.h_RL[::101, 0].copy()
    out["h_RL_center"] = fbp.h_RL[905:918, 0].copy()
    out["r"] = fbp.r.reshape(-1)[::4099].copy()
    out["phi"] = fbp.phi.reshape(-1)[::4099].copy()
    out["weight"] = (fbp.D * np.cos(fbp.nda))[::57].astype(np.float32)
    # ramp rows through the reference's conv_pj
    rows = (synth.hash_uniform((1, 6, 912), 51) * 4.0).astype(np.float32)
    pj_out = np.zeros_like(rows)
    out["ramp_rows"] = FB.conv_pj(pj_out, rows, fbp.h_RL, 6, 912, 1)
    # back-projection of scattered pixels through the reference's own (pure python here) fbp_cpu
    filt = (synth.hash_uniform((1, 2000, 912), 52) - 0.5).astype(np.float32)
    pix = np.array([0, 511, 512 * 511, 512 * 512 - 1, 256 * 512 + 256, 255 * 512 + 255, 100 * 512 + 400,
                    400 * 512 + 37, 17 * 512 + 300, 300 * 512 + 17, 256 * 512 + 3, 5 * 512 + 256], dtype=np.int32)
    vals = np.zeros(pix.size, dtype=np.float32)
    umap = np.zeros((20, pix.size))
    for q, p in enumerate(pix):
        i, j = divmod(int(p), 512)
        I1 = np.zeros((1, 1, 1), dtype=np.float32)
        FB.fbp_cpu(I1, 1, filt, fbp.phi[i:i + 1, j:j + 1], fbp.r[i:i + 1, j:j + 1], fbp.D, 1, fbp.M, fbp.N, fbp.theta,
                   fbp.da, fbp.nda)
        vals[q] = I1[0, 0, 0]
        for tt, t in enumerate(range(0, 2000, 100)):
            th = np.pi / 2 + (fbp.theta[t] - np.pi / 2) + fbp.phi[i, j]
            alpha = np.arctan(fbp.r[i, j] * np.sin(th) / (fbp.D + fbp.r[i, j] * np.cos(th)))
            umap[tt, q] = (alpha - fbp.nda[0]) / fbp.da + 0.5
    out["bp_pixels"] = pix
    out["bp_values"] = vals
    out["bp_umap"] = umap
    # the vectorised restatement must agree with the reference's sequential loop on those pixels
    sub = fbp_cpu_vectorised(fbp, filt)
    out["bp_vectorised_equal"] = np.array(np.array_equal(sub.reshape(-1)[pix], vals))
    # full convert() of a noisy phantom sinogram (vectorised back-projection patched in), sub-sampled 8x8
    FB.fbp_cpu = lambda I, BS, pj, phi, r, D, gridN, Mv, N, theta, da, nda: fbp_cpu_vectorised(fbp, pj)
    sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(3)), seed=3)
    img = fbp.convert(sino[None])
    out["convert_sub8"] = img[0, ::8, ::8].copy()
    out["convert_rows"] = img[0, 250:254, :].copy()
    save("fbp", **out)
    return img[0]


# ------------------------------------------------------------------ 8. curves / sharpen / units
def gen_misc():
    out = {}
    x = np.concatenate([np.linspace(0.2, 4.0, 200), [1.0, 1.7, 2.75, 0.999999, 1.7000001, 2.7500002]]).astype(np.float32)
    out["curve_x"] = x
    out["curve_img"] = U.curve_init()(x)
    out["curve_proj"] = U.proj_curv_init()(x)
    for name, c in (("img", U.curve_init()), ("proj", U.proj_curv_init())):
        out["coef_%s_p1" % name] = np.array(c.keywords["f1"].coeffs, dtype=np.float64)
        out["coef_%s_p2" % name] = np.array(c.keywords["f2"].coeffs, dtype=np.float64)
    img = torch.from_numpy(synth.hash_uniform((1, 1, 17, 13), 61))
    for n in (42, 70):
        out["sharpen_%d" % n] = U.tensor_sharpen(img, n).numpy()
    mu = torch.from_numpy(synth.hash_uniform((64,), 62) * 1.2 - 0.1)
    from Dataset.npz_data_loader import miu2pixel
    out["miu2pixel"] = miu2pixel(mu.clone()).numpy()
    # condition_lambda_ratio_cuda (Model/model.py:328-351): the reference's OWN kernel body executed per simulated
    # thread (ref_shim._PyFuncLaunch) at the launch configuration of its call site (:555-557), + the host-side clip
    lam = (synth.hash_uniform((2, 1, 9, 7), 63) * 3.9 + 0.01).astype(np.float32)      # Lambda(delta) in (0.01, 3.91)
    out["lambda_in"] = lam
    for (i, ts) in ((0, 15), (7, 15), (14, 15), (2, 3), (19, 20)):
        I = np.zeros_like(lam)
        M.condition_lambda_ratio_cuda[(64, 64, 1), (8, 8, 2)](I, np.array([0, i, i + 1]), 2, 9, 7, ts, lam)
        out["lambda_raw_i%d_ts%d" % (i, ts)] = I.copy()
        out["lambda_clip_i%d_ts%d" % (i, ts)] = np.clip(I, 0.05, 0.99)
        # the vectorised stand-in used for full-size runs must be the same function
        I2 = np.zeros_like(lam)
        ref_shim._lambda_ratio_numpy(I2, np.array([0, i, i + 1]), 2, 9, 7, ts, lam)
        assert np.array_equal(I, I2), "ref_shim numpy restatement != the reference kernel body"
    save("misc", **out)


# ------------------------------------------------------------------ 8b. the harness end to end
def gen_pipeline():
    """Drives the reference's own progressive_domain_denoiser.progressive_denoiser() (Utils/train_test_utils.py:552-567:
    proj_denoiser -> convertor (FBP, CPU path) -> tensor_sharpen -> img_denoiser -> ultra) on one synthetic low-dose
    sinogram at the true geometry with reduced UNets, hashed noise in place of torch.randn_like, and records
    sub-sampled outputs.  Stubs: dataset/dataloader init (no dataset offline), the pure-python back-projection loop
    is replaced by its vectorised twin (bit-equal on the pixels gen_fbp() checks)."""
    import argparse
    import tempfile
    from tests.golden.cases import PIPE_OPT, PIPE_SEEDS, PIPE_SHARPEN
    from Config.default_config import default_cfg, cfg_load
    import json as _json
    cwd = os.getcwd()
    os.chdir(ref_shim.REFERENCE_ROOT)                 # init_convertor reads Recon/Simens_*.txt relative to the cwd
    orig_randn, orig_fbp_cpu, orig_loader = torch.randn_like, FB.fbp_cpu, U.progressive_domain_denoiser.init_data_loader
    out = {}
    try:
        argv, sys.argv = sys.argv, sys.argv[:1]
        opt = default_cfg()
        sys.argv = argv
        with open(os.path.join(ref_shim.REFERENCE_ROOT, "Config", "Mayo-Config", "test_progressive_option.json")) as f:
            cfg_load(_json.load(f), opt.__dict__)
        cfg_load(dict(PIPE_OPT, device="cpu"), opt.__dict__)
        U.progressive_domain_denoiser.init_data_loader = lambda self: None
        tmp = tempfile.mkdtemp(prefix="ipdm_golden_")
        den = U.progressive_domain_denoiser(opt, result_save_path=tmp)
        for net, seed in ((den.proj_model, PIPE_SEEDS["proj_weights"]), (den.img_model, PIPE_SEEDS["img_weights"])):
            shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
            net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=seed).items()})
            net.eval()
        out["proj_keys"] = np.array(list(den.proj_model.state_dict().keys()))
        out["img_keys"] = np.array(list(den.img_model.state_dict().keys()))
        fbp_obj = den.convertor.__self__
        FB.fbp_cpu = lambda I, BS, pj, phi, r, D, gridN, Mv, N, theta, da, nda: fbp_cpu_vectorised(fbp_obj, pj)
        sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(PIPE_SEEDS["phantom"])), seed=PIPE_SEEDS["dose"])
        den.data_sample_load(ldct=None, ldproj=torch.from_numpy(sino)[None, None], fdproj=None, fdct=None)
        feed = _NoiseFeed(PIPE_SEEDS["noise"])
        torch.randn_like = feed
        res = den.progressive_denoiser(sharpen_num=PIPE_SHARPEN)
        out["a_ndraws"] = np.array(feed.k)
        out["a_final_sub4"] = res.numpy()[0, 0, ::4, ::4].copy()
        out["a_final_rows"] = res.numpy()[0, 0, 254:258].copy()
        out["a_noise_strength"] = np.array(str(den.noise_strength))
        out["a_conv_keys"] = np.array(sorted(den.proj_denoise_convert2img_result.keys()))
        out["a_prog_keys"] = np.array(sorted(den.progressive_denoise_result.keys()))
        out["a_convert_sub8"] = den.proj_denoise_convert2img_result[-1][0, 0, ::8, ::8].copy()
        out["a_prog_last_sub8"] = den.progressive_denoise_result[-1][0, 0, ::8, ::8].copy()
        from Dataset.npz_data_loader import miu2pixel
        ph = miu2pixel(synth.rasterize(synth.ellipse_phantom(PIPE_SEEDS["phantom"])))
        mse = float(np.mean((miu2pixel(res.numpy()[0, 0]) - ph) ** 2))
        out["a_psnr_vs_phantom"] = np.array(10 * math.log10(1.0 / mse))
        print("  pipeline A: draws %d  PSNR vs phantom %.3f dB  noise_strength %s" % (feed.k, out["a_psnr_vs_phantom"], den.noise_strength))
        # run B: every intermediate kept (save_it_state_* = True, save_proj_state=True): result-dict layout + iterates
        den.temp_clear()
        den.update_opt(dict(save_it_state_proj=True, save_it_state_img=True))
        feed = _NoiseFeed(PIPE_SEEDS["noise"])
        torch.randn_like = feed
        res_b = den.progressive_denoiser(save_proj_state=True, sharpen_num=PIPE_SHARPEN)
        out["b_equals_a"] = np.array(bool(np.array_equal(res_b.numpy(), res.numpy())))
        for name in ("proj_denoise_result", "proj_denoise_convert2img_result", "progressive_denoise_result",
                     "img_denoise_result"):
            d = getattr(den, name)
            out["b_%s_keys" % name] = np.array(sorted(d.keys()))
            for k in sorted(d.keys()):
                a = d[k]
                out["b_%s_%s" % (name, k)] = a[0, 0, ::32, ::16].copy() if a.shape[-2] > 512 else a[0, 0, ::16, ::16].copy()
        den.reset_opt()
    finally:
        torch.randn_like, FB.fbp_cpu = orig_randn, orig_fbp_cpu
        U.progressive_domain_denoiser.init_data_loader = orig_loader
        os.chdir(cwd)
    save("pipeline", **out)


def _ref_denoiser(opt_over):
    """The reference's progressive_domain_denoiser on CPU with the PIPE_OPT networks (dataset init stubbed)."""
    import tempfile
    import json as _json
    from tests.golden.cases import PIPE_OPT, PIPE_SEEDS
    from Config.default_config import default_cfg, cfg_load
    argv, sys.argv = sys.argv, sys.argv[:1]
    opt = default_cfg()
    sys.argv = argv
    with open(os.path.join(ref_shim.REFERENCE_ROOT, "Config", "Mayo-Config", "test_progressive_option.json")) as f:
        cfg_load(_json.load(f), opt.__dict__)
    cfg_load(dict(PIPE_OPT, device="cpu", **opt_over), opt.__dict__)
    U.progressive_domain_denoiser.init_data_loader = lambda self: None
    den = U.progressive_domain_denoiser(opt, result_save_path=tempfile.mkdtemp(prefix="ipdm_golden_"))
    for net, seed in ((den.proj_model, PIPE_SEEDS["proj_weights"]), (den.img_model, PIPE_SEEDS["img_weights"])):
        if net is None:
            continue
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=seed).items()})
        net.eval()
    return den


def gen_pipeline_img():
    """img_denoiser(mode="img_only") of the reference harness (Utils/train_test_utils.py:482-550: img loop + ultra) on an
    input that is regenerable from integer hashes (rasterised phantom + hashed noise), so that the oracle and the HIP
    path start from bit-identical data: pins the image-domain half of the orchestration without the FBP image's
    summation-order differences in front of 17 network evaluations."""
    from tests.golden.cases import PIPE_SEEDS
    cwd = os.getcwd()
    os.chdir(ref_shim.REFERENCE_ROOT)
    orig_randn, orig_loader = torch.randn_like, U.progressive_domain_denoiser.init_data_loader
    out = {}
    try:
        den = _ref_denoiser(dict(mode="test_img", save_it_state_img=True))
        x = synth.rasterize(synth.ellipse_phantom(PIPE_SEEDS["phantom"])) + 0.004 * synth.hash_normal((512, 512), 74)
        feed = _NoiseFeed(PIPE_SEEDS["noise"] + 2)
        torch.randn_like = feed
        res = den.img_denoiser(torch.from_numpy(x.astype(np.float32))[None, None], noise_strength=None, mode="img_only")
        out["ndraws"] = np.array(feed.k)
        out["final_sub4"] = res.numpy()[0, 0, ::4, ::4].copy()
        out["final_rows"] = res.numpy()[0, 0, 254:258].copy()
        out["keys"] = np.array(sorted(den.img_denoise_result.keys()))
        for k in sorted(den.img_denoise_result.keys()):
            out["img_" + k] = den.img_denoise_result[k][0, 0, ::16, ::16].copy()
        assert len(den.progressive_denoise_result) == 0
    finally:
        torch.randn_like = orig_randn
        U.progressive_domain_denoiser.init_data_loader = orig_loader
        os.chdir(cwd)
    save("pipeline_img", **out)


def gen_pipeline_c1():
    """BASELINE.json config C1 at its literal setting: ONE 512x512 slice, image domain only, t_start_img=[5],
    constant_guidance_img=0.45, no ultra pass -- the reference harness's img_denoiser(mode="img_only")
    (Utils/train_test_utils.py:482-550) on an input regenerable from integer hashes, every stored iterate."""
    from tests.golden.cases import PIPE_SEEDS, C1_OPT, C1_NOISE_SEED, C1_INPUT_SEED
    cwd = os.getcwd()
    os.chdir(ref_shim.REFERENCE_ROOT)
    orig_randn, orig_loader = torch.randn_like, U.progressive_domain_denoiser.init_data_loader
    out = {}
    try:
        den = _ref_denoiser(dict(C1_OPT, mode="test_img"))
        x = synth.rasterize(synth.ellipse_phantom(PIPE_SEEDS["phantom"])) + 0.004 * synth.hash_normal((512, 512), C1_INPUT_SEED)
        feed = _NoiseFeed(C1_NOISE_SEED)
        torch.randn_like = feed
        res = den.img_denoiser(torch.from_numpy(x.astype(np.float32))[None, None], noise_strength=None, mode="img_only")
        out["ndraws"] = np.array(feed.k)
        out["final_sub4"] = res.numpy()[0, 0, ::4, ::4].copy()
        out["final_rows"] = res.numpy()[0, 0, 254:258].copy()
        out["keys"] = np.array(sorted(den.img_denoise_result.keys()))
        for k in sorted(den.img_denoise_result.keys()):
            out["img_" + k] = den.img_denoise_result[k][0, 0, ::16, ::16].copy()
        from Dataset.npz_data_loader import miu2pixel
        ph = miu2pixel(synth.rasterize(synth.ellipse_phantom(PIPE_SEEDS["phantom"])))
        mse = float(np.mean((miu2pixel(res.numpy()[0, 0]) - ph) ** 2))
        out["psnr_vs_phantom"] = np.array(10.0 * np.log10(1.0 / mse))
        assert len(den.progressive_denoise_result) == 0
    finally:
        torch.randn_like = orig_randn
        U.progressive_domain_denoiser.init_data_loader = orig_loader
        os.chdir(cwd)
    save("pipeline_c1", **out)


# ------------------------------------------------------------------ 9. ART data tables
def gen_art_tables():
    """Samples of the two data files the ART convertor is driven by (Recon/Simens_alut.txt: the pixel-area table
    [181, 1501]; Recon/Simens_theta.txt: the 2000 view angles): every 7th angle row x every 13th distance column plus the
    first / last rows and columns, and every 9th view angle."""
    ref = os.path.join(os.path.dirname(os.path.dirname(M.__file__)), "Recon")
    sa = np.fromfile(os.path.join(ref, "Simens_alut.txt"), "float32").reshape(181, 1501)
    st = np.fromfile(os.path.join(ref, "Simens_theta.txt"), "float32")
    rows = np.unique(np.concatenate([np.arange(0, 181, 7), [1, 179, 180]]))
    cols = np.unique(np.concatenate([np.arange(0, 1501, 13), [1, 2, 1498, 1499, 1500]]))
    save("art_tables", rows=rows, cols=cols, lut=sa[np.ix_(rows, cols)], lut_shape=np.array(sa.shape),
         lut_sum=np.array(sa.astype(np.float64).sum()), theta_idx=np.arange(0, 2000, 9), theta=st[::9],
         theta_n=np.array(st.size))


# ------------------------------------------------------------------ 10. NQM (the reference's own metric code)
def gen_metrics():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_nqm", os.path.join(os.path.dirname(os.path.dirname(M.__file__)),
                                                                         "Utils", "NQM.py"))
    nqm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nqm)
    out = {}
    for tag, (n, seed, noise) in {"a": (128, 81, 0.02), "b": (96, 82, 0.08), "c": (512, 83, 0.01)}.items():
        yy, xx = np.mgrid[0:n, 0:n] / float(n)
        ref = (0.5 + 0.3 * np.sin(9 * xx) * np.cos(7 * yy) + 0.15 * (((xx - .4) ** 2 + (yy - .6) ** 2) < .04)).astype(np.float32)
        qry = (ref + noise * synth.hash_normal((n, n), seed)).astype(np.float32)
        out["nqm_" + tag] = np.array(float(nqm.NQM(ref, qry)))
    save("metrics", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:            # python tests/golden/make_golden.py pipeline misc ...
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_schedule()
    gen_groups()
    gen_unet()
    gen_ops()
    gen_step()
    gen_loops()
    gen_adaptive()
    gen_sparse()
    gen_misc()
    gen_pipeline()
    gen_pipeline_img()
    gen_pipeline_c1()
    gen_fbp()
    gen_art_tables()
    gen_metrics()
