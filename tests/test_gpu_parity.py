"""GPU parity tests: the HIP path (through the C ABI of libipdm_hip.so) against the CPU oracle and
the golden vectors of the imported reference.  Tolerances are absolute unless noted and written
next to each check; north_star: 1e-5 max-abs on the FBP index map, 1e-4 relative PSNR end to end."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import diffusion as od   # noqa: E402
from oracle import fbp as of         # noqa: E402
from oracle import unet as ou        # noqa: E402
from ipdm_pytorch_amd import synth   # noqa: E402
from tests._oracle_pool import host_threads   # noqa: E402
from tests.golden.cases import SMALL_CFGS, SMALL_SHAPES, LOOP_CFG, LOOP_CASES, noise_feed  # noqa: E402

DEV = "cuda:0"

# End-to-end max-abs budget of two float32 evaluations of the REDUCED (smoke) pipeline, whose random-weight networks amplify
# float32 rounding ~100x in one of the image-domain passes ("the arbiter"): the CPU oracle itself sits
# 2.8e-4 / 0.83 = 3.4e-4 (relative to the output scale; max over 262k pixels, rms 8e-6) from the float64 value of the same
# function, this library 3.8e-4 / 0.83 (test_smoke_pipeline_fp64_arbiter measures both live and bounds the library's by 1.5x
# the oracle's).  Two such evaluations may differ by up to the sum; a bound below the oracle's own distance to the truth
# (the 2e-4 of round 1) tests luck, not the library.  Measured |hip - oracle|: 1.4e-4 .. 2.8e-4 across builds.
E2E_MAX_REL = 6e-4
# The production networks at full size do not amplify like that: 1.3e-6 .. 1.9e-5 measured (5 seeds x few steps; the
# headline length).  PSNR, north_star's criterion, is asserted at 1e-4 relative everywhere.
FULL_SIZE_MAX_REL = 1e-4


def _dev(a):
    return torch.as_tensor(a).to(DEV)


# =========================================================================== FBP
@pytest.fixture(scope="module")
def fbp():
    from ipdm_pytorch_amd.fbp import FBP
    return FBP(DEV)


def test_fbp_tables_match_reference_geometry(fbp, golden):
    g = golden("fbp")
    np.testing.assert_array_equal(fbp.table(0)[::97], g["theta"])
    np.testing.assert_array_equal(fbp.table(3)[::57], g["nda"])
    np.testing.assert_array_equal(fbp.table(4)[::101], g["h_RL"])
    np.testing.assert_array_equal(fbp.table(4)[905:918], g["h_RL_center"])
    np.testing.assert_array_equal(fbp.table(2)[::4099], g["r"])
    assert np.abs(fbp.table(1)[::4099] - g["phi"]).max() < 1e-15 * 7     # libm atan vs numpy: <= 1 ulp
    assert np.abs(fbp.table(5)[::57] - g["weight"]).max() <= 8e-6        # cosf vs np.cos(f32): 1 ulp of ~59


def test_fbp_ramp_rows_golden(fbp, golden):
    g = golden("fbp")
    rows = (synth.hash_uniform((1, 6, 912), 51) * 4.0).astype(np.float32)
    # the plan has 2000 views; filter a [1,2000,912] sinogram whose first 6 rows are the golden rows.
    # golden rows went through conv_pj directly (no flip/weight) -> undo weight with flip=False path:
    sino = np.zeros((1, 2000, 912), dtype=np.float32)
    sino[0, :6] = rows
    geo = of.FBPGeometry()
    got = fbp.filter_device(_dev(sino), flip=False).cpu().numpy()[0, :6]
    want = of.ramp_filter(_geo_views(geo, 6), of.weight_sinogram(geo, rows, flip=False))[0]
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-6 * scale


def _geo_views(geo, n):
    import copy
    g2 = copy.copy(geo)
    g2.n_views = n
    return g2


def test_fbp_filter_full(fbp):
    geo = of.FBPGeometry()
    sino = (synth.hash_uniform((2, 2000, 912), 71) * 6.0).astype(np.float32)
    got = fbp.filter_device(_dev(sino), flip=True).cpu().numpy()
    want = of.ramp_filter(geo, of.weight_sinogram(geo, sino, flip=True))
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


def test_fbp_index_map(fbp, golden):
    """north_star: <= 1e-5 max-abs on the FBP index map (detector coordinate u)."""
    g = golden("fbp")
    geo = of.FBPGeometry()
    rng = np.random.default_rng(5)
    pix = np.concatenate([g["bp_pixels"], rng.integers(0, 512 * 512, 500).astype(np.int32)])
    u_dev = fbp.index_map(pix).cpu().numpy()
    _, u_ref = of.backproject(geo, np.zeros((1, 2000, 912), np.float32), pixels=pix, want_umap=True)
    assert np.abs(u_dev - u_ref).max() < 1e-9
    assert np.abs(u_dev[::100, :12] - g["bp_umap"]).max() < 1e-9        # reference's own expression
    assert np.array_equal(np.floor(u_dev), np.floor(u_ref)) or (np.floor(u_dev) != np.floor(u_ref)).mean() < 1e-6


def test_fbp_backprojection_full_image(fbp, golden):
    g = golden("fbp")
    geo = of.FBPGeometry()
    filt = (synth.hash_uniform((1, 2000, 912), 52) - 0.5).astype(np.float32)
    got = fbp.backproject_device(_dev(filt)).cpu().numpy()
    want = of.backproject(geo, filt)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-6 * scale
    # the reference's own sequential loop on scattered pixels
    assert np.abs(got.reshape(-1)[g["bp_pixels"]] - g["bp_values"]).max() <= 2e-6 * scale


def test_fbp_convert_phantom_batch(fbp, golden):
    g = golden("fbp")
    sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(3)), seed=3)
    batch = np.stack([sino, sino * 0.5, sino[::-1].copy()])          # ragged batch of 3 (KB=2 + KB=1 launches)
    got = fbp.convert_device(_dev(batch)).cpu().numpy()
    scale = np.abs(g["convert_rows"]).max()
    assert np.abs(got[0, ::8, ::8] - g["convert_sub8"]).max() <= 5e-6 * scale
    assert np.abs(got[0, 250:254] - g["convert_rows"]).max() <= 5e-6 * scale
    # linearity (size-independent property): FBP(0.5 s) == 0.5 FBP(s) exactly (power-of-two scaling)
    np.testing.assert_array_equal(got[1], 0.5 * got[0])
    want2 = of.convert(of.FBPGeometry(), batch[2:3])
    assert np.abs(got[2] - want2[0]).max() <= 5e-6 * np.abs(want2).max()
    # reference return type: Tensor in -> CPU Tensor out
    out = fbp.convert(torch.from_numpy(sino))
    assert isinstance(out, torch.Tensor) and out.device.type == "cpu" and tuple(out.shape) == (1, 512, 512)


def test_sharpen(golden):
    from ipdm_pytorch_amd.fbp import tensor_sharpen
    g = golden("misc")
    img = torch.from_numpy(synth.hash_uniform((1, 1, 17, 13), 61))
    for n in (42, 70):
        got = tensor_sharpen(img.to(DEV), n).cpu().numpy()
        np.testing.assert_allclose(got, g["sharpen_%d" % n], rtol=0, atol=1e-6)
    assert tensor_sharpen(img.to(DEV), -1).data_ptr() != 0


# =========================================================================== diffusion arithmetic
@pytest.fixture(scope="module")
def gd5():
    from ipdm_pytorch_amd.diffusion import GaussianDiffusion
    return GaussianDiffusion(1000, "cosine", 5)


def test_schedule_tables_golden(golden):
    from ipdm_pytorch_amd.diffusion import GaussianDiffusion, cosine_lambda
    g = golden("schedule")
    for p in (1, 5):
        gd = GaussianDiffusion(1000, "cosine", p)
        tab = np.array([gd.coeffs(t) for t in range(30)], dtype=np.float32).T
        # float64 tables computed with libm instead of torch: identical after the cast except for rare 1-ulp flips
        np.testing.assert_allclose(tab, g["tables_p%d" % p], rtol=2e-7, atol=0)
    for ts, power in ((15, 1), (15, 10), (5, 10), (20, 1)):
        got = np.array([cosine_lambda(ts, power, i) for i in range(ts)])
        np.testing.assert_allclose(got, g["lambda_ts%d_p%d" % (ts, power)], rtol=1e-13, atol=0)


def test_single_step_golden(gd5, golden):
    g = golden("step")
    shape = (1, 1, 16, 12)
    x_t = torch.from_numpy(synth.hash_normal(shape, 31)) * 0.3 + 0.5
    x_0 = torch.from_numpy(synth.hash_normal(shape, 32)) * 0.2 + 0.5
    pred = torch.from_numpy(synth.hash_normal(shape, 33)) * 1.7 + 0.1
    lam_small = torch.from_numpy(synth.hash_uniform((1, 1, 4, 3), 34)) * 0.9 + 0.05
    noise = torch.from_numpy(synth.hash_normal(shape, 35 * 1000))
    cases = {"scalar_t7": (7, 0.45, True), "scalar_t0": (0, 0.45, True),
             "tensor0d_t5": (5, float(od.cosine_beta_schedule(15, schedule_power=1)[5]), False),
             "map_t3": (3, lam_small.to(DEV), False)}
    for tag, (t, lam, clip) in cases.items():
        got = gd5.p_sample_condition(None, x_t.to(DEV), x_0.to(DEV), t, clip_denoised=clip, lambda_=lam,
                                     noise=noise.to(DEV), eps_pred=pred.to(DEV)).cpu().numpy()
        np.testing.assert_allclose(got, g[tag], rtol=0, atol=2e-6)


def test_step_full_size_per_slice(gd5):
    """B=3 sinogram-sized slices: slice b must equal the oracle run on that slice alone."""
    shape = (3, 1, 2000, 912)
    x_t = torch.from_numpy(synth.hash_normal(shape, 81)) * 0.4 + 1.5
    x_0 = torch.from_numpy(synth.hash_normal(shape, 82)) * 0.3 + 1.5
    pred = torch.from_numpy(synth.hash_normal(shape, 83)) * torch.tensor([1.0, 2.5, 0.3]).view(3, 1, 1, 1) + 0.2
    noise = torch.from_numpy(synth.hash_normal(shape, 84))
    lam_small = torch.from_numpy(synth.hash_uniform((3, 1, 500, 228), 85)) * 0.9 + 0.05
    sch = od.Schedule(1000, 5)
    for lam_dev, lam_cpu in ((0.37, 0.37), (lam_small.to(DEV), lam_small)):
        got = gd5.p_sample_condition(None, x_t.to(DEV), x_0.to(DEV), 9, clip_denoised=False, lambda_=lam_dev,
                                     noise=noise.to(DEV), eps_pred=pred.to(DEV)).cpu()
        for b in range(3):
            lam_b = lam_cpu if not isinstance(lam_cpu, torch.Tensor) else \
                torch.nn.functional.interpolate(lam_cpu[b:b + 1], size=(2000, 912), mode="nearest")
            want = od.p_sample_condition(sch, lambda x, t: pred[b:b + 1], x_t[b:b + 1], x_0[b:b + 1], 9, lam_b, False,
                                         noise[b:b + 1])
            assert (got[b:b + 1] - want).abs().max() <= 5e-6 * want.abs().max()


def test_q_sample_and_elementwise(gd5):
    from ipdm_pytorch_amd import _lib
    shape = (2, 1, 64, 48)
    x = torch.from_numpy(synth.hash_normal(shape, 91))
    z = torch.from_numpy(synth.hash_normal(shape, 92))
    sch = od.Schedule(1000, 5)
    got = gd5.q_sample(x.to(DEV), 15, z.to(DEV)).cpu()
    want = od.q_sample(sch, x, 15, z)
    assert (got - want).abs().max() <= 1e-6
    y = torch.empty_like(x, device=DEV)
    for mode in (0, 1):
        _lib.call("ipdm_clamp", _lib.ptr(x.to(DEV)), _lib.ptr(y), x.numel(), mode, _lib.current_stream())
        want = x.clamp(0, 1) if mode == 0 else x.clamp(min=0)
        assert torch.equal(y.cpu(), want)


def test_randn_statistics_and_shard_invariance():
    from ipdm_pytorch_amd.diffusion import NoiseSource
    x = torch.empty((4, 1, 512, 512), device=DEV)
    a = NoiseSource(seed=7, slice_id0=0).next_like(x)
    b = NoiseSource(seed=7, slice_id0=2).next_like(x[:2])
    assert torch.equal(a[2:], b)                         # slice 2,3 identical whichever shard generates them
    s2 = NoiseSource(seed=7, slice_id0=0)
    s2.next_like(x)
    c = s2.next_like(x)
    assert not torch.equal(a, c)                          # next draw differs
    v = a.double().cpu().numpy().reshape(4, -1)
    assert np.abs(v.mean(axis=1)).max() < 0.01 and np.abs(v.std(axis=1) - 1).max() < 0.01
    assert abs(np.corrcoef(v[0], v[1])[0, 1]) < 0.01
    k = ((v ** 4).mean() / (v ** 2).mean() ** 2)
    assert abs(k - 3.0) < 0.05                            # kurtosis of a Gaussian
    odd = torch.empty((1, 1, 7, 5), device=DEV)
    assert torch.isfinite(NoiseSource(1).next_like(odd)).all()


def test_randn_against_the_cpu_restatement():
    """ipdm_randn against oracle/noise.py (Philox4x32-10 keyed by seed, GLOBAL slice id, draw index, element quad; Box-Muller in
    float32): the key layout is exact -- one wrong word of the counter gives unrelated values --, the float32 transform agrees to
    a few ulps of logf / sincosf (4e-6 relative to max(1, |z|)).  Cases: the two tensor shapes of the path, a count that is not
    a multiple of 4, slice ids and draw indices beyond 32 bits, a 64-bit seed."""
    from ipdm_pytorch_amd.diffusion import NoiseSource
    from oracle import noise
    for seed, slice0, draw, shape in ((7, 0, 0, (2, 1, 512, 512)), (1234, 5, 44, (2, 1, 2000, 912)), (3, 1, 2, (3, 1, 7, 5)),
                                      ((1 << 40) + 17, (1 << 33) + 2, (1 << 32) + 9, (2, 1, 64, 48))):
        src = NoiseSource(seed=seed, slice_id0=slice0)
        src.draw = draw
        got = src.next_like(torch.empty(shape, device=DEV)).cpu().numpy()
        n = int(np.prod(shape[1:]))
        for b in range(shape[0]):
            want = noise.randn(seed, slice0 + b, draw, n)
            err = np.abs(got[b].reshape(-1) - want) / np.maximum(1.0, np.abs(want))
            assert err.max() <= 4e-6, (seed, slice0 + b, draw, float(err.max()))


def test_slice_median():
    from ipdm_pytorch_amd import _lib
    for n in (1, 2, 7, 1000, 1824000, 262144):
        x = torch.from_numpy(synth.hash_normal((3, n), 100 + n % 97))
        if n > 100:
            x[1, : n // 2] = 0.25      # heavy ties
            x[2] = -x[2].abs()
        xd = x.to(DEV)
        med = torch.empty(3, device=DEV)
        ws = torch.empty(1 << 16, dtype=torch.uint8, device=DEV)
        _lib.call("ipdm_slice_median", _lib.ptr(xd), _lib.ptr(med), 3, n, _lib.ptr(ws), ws.numel(), _lib.current_stream())
        want = torch.stack([torch.median(x[b]) for b in range(3)])
        assert torch.equal(med.cpu(), want), (n, med.cpu(), want)


def test_guidance_map_and_lambda_ratio(gd5):
    for mode, shape, amp in (("proj", (2, 1, 2000, 912), 7.0), ("img", (2, 1, 512, 512), 30.0),
                             ("proj", (1, 1, 40, 24), 7.0)):
        if mode == "proj":
            img = torch.from_numpy(synth.hash_uniform(shape, 111)) * 4.0
            x = img + torch.from_numpy(synth.hash_normal(shape, 112)) * 0.08
        else:
            img = torch.from_numpy(synth.hash_uniform(shape, 113)) * 0.05 + 0.17
            x = img + torch.from_numpy(synth.hash_normal(shape, 114)) * 0.004
        Lam, emax = gd5.guidance_map(x.to(DEV), img.to(DEV), mode, 4, amp)
        for b in range(shape[0]):
            e, want = od.delta_map(x[b:b + 1], img[b:b + 1], mode, 4, amp)
            # exp(amp*delta) amplifies the 1e-7 pooling differences; the curve has slope up to ~40
            assert (Lam[b:b + 1].cpu() - want).abs().max() <= 2e-4, (mode, b)
            assert abs(float(emax[b]) - float(e.max())) <= 1e-5 * float(e.max())
        for (i, ts) in ((14, 15), (0, 15), (2, 3)):
            got = gd5.lambda_ratio(Lam, i, ts).cpu()
            want = od.lambda_ratio_map(Lam.cpu(), i, ts)
            assert (got - want).abs().max() <= 1e-6


# =========================================================================== conv / attention kernels
def _conv_case(B, C1, C2, Hs, Ws, H, W, Cout, ks, stride, act, res, seed):
    from ipdm_pytorch_amd import _lib
    import torch.nn.functional as F
    Cin = C1 + C2
    x1 = torch.from_numpy(synth.hash_normal((B, C1, Hs, Ws), seed))
    x2 = torch.from_numpy(synth.hash_normal((B, C2, Hs, Ws), seed + 1)) * 2 + 0.5 if C2 else None
    w = torch.from_numpy(synth.hash_normal((Cout, Cin, ks, ks), seed + 2)) / np.sqrt(Cin * ks * ks)
    bias = torch.from_numpy(synth.hash_normal((Cout,), seed + 3))
    gamma = torch.from_numpy(synth.hash_uniform((Cin,), seed + 4)) + 0.5
    beta = torch.from_numpy(synth.hash_normal((Cin,), seed + 5)) * 0.2
    groups = ou.gn_groups(Cin)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    h = xin
    if act:
        h = F.group_norm(h, groups, gamma, beta, eps=1e-5)
        if act == 2:
            h = F.silu(h)
    if (H, W) != (Hs, Ws):
        h = F.interpolate(h, size=(H, W), mode="nearest")
    want = F.conv2d(h, w, bias, stride=stride, padding=ks // 2)
    r = torch.from_numpy(synth.hash_normal(tuple(want.shape), seed + 6)) if res else None
    if res:
        want = want + r
    out = torch.empty(tuple(want.shape), device=DEV)
    x1d = x1.to(DEV)
    x2d = x2.to(DEV) if x2 is not None else None
    rd = r.to(DEV) if res else None
    wn, bn, gn_, ben = (np.ascontiguousarray(t.numpy()) for t in (w, bias, gamma, beta))
    _lib.call("ipdm_op_conv2d", _lib.ptr(x1d), C1, _lib.ptr(x2d), C2, B, Hs, Ws, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, ks,
              stride, act, groups, _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
    got = out.cpu()
    err = (got - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), (err, (B, C1, C2, Hs, Ws, H, W, Cout, ks, stride, act, res))
    return got


def _conv_case_out(case, seed):
    return _conv_case(*case, seed=seed)


# which kernel the cases that are meant to cover a specific one must land on (ipdm_conv_kernel_code; plain single-source shapes only)
CONV_CASE_KERNELS = {
    (8, 128, 0, 64, 96, 64, 96, 128, 3, 1, 2, True): 2,        # conv_wino2
    (4, 128, 0, 200, 96, 200, 96, 128, 3, 1, 2, True): 2,
    (2, 128, 0, 19, 250, 19, 250, 128, 3, 1, 2, True): 1,       # 32 of the 128-cout tiles: the 64-cout Winograd kernel
    (2, 256, 0, 13, 125, 13, 125, 256, 3, 1, 2, True): 9,       # 16 direct tiles per sample: K slices inside conv_wino2
    (1, 256, 0, 32, 32, 32, 32, 256, 3, 1, 2, True): 9,
    (2, 64, 0, 32, 32, 32, 32, 64, 3, 1, 2, True): 4,           # 64 couts, few tiles: K-split direct kernel
    (1, 256, 0, 8, 8, 8, 8, 768, 1, 1, 1, False): 4,
    (2, 256, 0, 17, 57, 17, 57, 256, 1, 1, 0, True): 4,
    (1, 8, 4, 33, 57, 33, 57, 8, 3, 1, 2, False): 5,            # (code of the same shape with ONE source)
    (1, 64, 0, 70, 100, 70, 100, 1, 3, 1, 2, False): 5,
    (2, 64, 0, 31, 45, 31, 45, 64, 3, 2, 0, False): 4,          # stride 2 to 16x23: conv_ws with a K split
    (2, 64, 0, 37, 117, 19, 59, 128, 3, 2, 0, False): 4,
    (2, 256, 0, 57, 125, 57, 125, 768, 1, 1, 1, False): 10,    # conv_pw: the barrier-free pointwise kernel
    (2, 256, 0, 40, 72, 40, 72, 256, 1, 1, 0, True): 10,
    (2, 128, 128, 45, 95, 45, 95, 128, 1, 1, 0, False): 10,
    (1, 64, 0, 64, 64, 64, 64, 128, 1, 1, 0, False): 10,
}


@pytest.mark.parametrize("case", [
    # B, C1, C2, Hs, Ws, H, W, Cout, ks, stride, act, res
    (1, 1, 0, 24, 40, 24, 40, 64, 3, 1, 0, False),          # stem
    (2, 64, 0, 32, 32, 32, 32, 64, 3, 1, 2, True),          # resblock conv2 + residual
    (1, 128, 64, 16, 32, 16, 32, 64, 3, 1, 2, False),       # concat input, GN straddling the two sources (32 groups of 6)
    (1, 8, 4, 33, 57, 33, 57, 8, 3, 1, 2, False),           # tiny channels, odd sizes (proj UNet level 0)
    (1, 144, 0, 9, 7, 9, 7, 24, 3, 1, 2, False),            # 36 groups
    (2, 64, 0, 31, 45, 31, 45, 64, 3, 2, 0, False),         # down-sample, odd input
    (1, 16, 0, 29, 63, 57, 125, 16, 3, 1, 0, False),        # nearest up-sample to an explicit odd size
    (1, 256, 0, 8, 8, 8, 8, 768, 1, 1, 1, False),           # qkv 1x1 with GN (no SiLU)
    (2, 192, 0, 10, 12, 10, 12, 128, 1, 1, 0, True),        # shortcut 1x1 + residual
    (1, 64, 0, 70, 100, 70, 100, 1, 3, 1, 2, False),        # out conv, Cout=1
    (1, 130, 0, 12, 33, 12, 33, 70, 3, 1, 0, False),        # channels not multiples of the tile sizes
    (1, 8, 4, 33, 57, 33, 57, 8, 1, 1, 0, False),           # narrow 1x1 shortcut over a concat (direct kernel)
    (2, 16, 0, 20, 70, 20, 70, 16, 1, 1, 0, True),          # narrow 1x1 + residual, width not a multiple of 4
    (1, 144, 0, 24, 40, 24, 40, 16, 3, 1, 2, False),        # 144 -> 16 up-block conv (direct kernel, 18 channel chunks)
    (8, 128, 0, 64, 96, 64, 96, 128, 3, 1, 2, True),        # enough tiles for the 8x32x128 persistent schedule (several rounds)
    (2, 64, 64, 48, 40, 48, 40, 64, 3, 1, 2, True),         # 16x32x64 tiles, concat, ragged right edge
    (1, 256, 0, 32, 32, 32, 32, 256, 3, 1, 2, True),        # batch 1, 32 tiles: K split into 8 slices + combine (bias, residual)
    (1, 256, 256, 16, 24, 16, 24, 256, 3, 1, 2, False),     # K split over a concat input
    (1, 256, 0, 16, 16, 16, 16, 768, 1, 1, 1, False),       # 1x1 (qkv) with GN, K split into 4
    (1, 256, 0, 31, 45, 31, 45, 256, 3, 2, 0, False),       # stride 2, K split
    (1, 192, 0, 29, 63, 29, 63, 72, 3, 1, 2, True),         # K split with ragged couts and width % 4 != 0
    (2, 128, 0, 19, 250, 19, 250, 128, 3, 1, 2, True),      # width % 4 == 2: 16-byte epilogue with a ragged last run (2 pixels)
    (2, 256, 0, 13, 125, 13, 125, 256, 3, 1, 2, True),      # width % 4 == 1, residual on the partial run
    (2, 128, 0, 21, 63, 21, 63, 128, 3, 1, 0, False),       # width % 4 == 3, 63 columns = one full + one ragged tile column
    (2, 256, 0, 17, 57, 17, 57, 256, 1, 1, 0, True),        # 1x1 with residual, width % 4 == 1
    (2, 64, 0, 37, 117, 19, 59, 128, 3, 2, 0, False),       # stride 2 to 19x59
    (4, 128, 0, 200, 96, 200, 96, 128, 3, 1, 2, True),      # 300 tiles of 8x32x128: one full round of the persistent schedule + a 44-tile tail
    (2, 128, 64, 203, 90, 203, 90, 256, 3, 1, 2, False),    # two rounds with ragged rows (203), ragged width (90), concat, two cout tiles
    (2, 256, 0, 57, 125, 57, 125, 768, 1, 1, 1, False),     # pointwise kernel: qkv with GroupNorm at the proj UNet's T = 7125 (ragged last item: 21 pixels)
    (2, 256, 0, 40, 72, 40, 72, 256, 1, 1, 0, True),        # ... proj_out with its residual, whole items
    (2, 128, 128, 45, 95, 45, 95, 128, 1, 1, 0, False),     # ... shortcut over a concat, ragged
    (1, 64, 0, 64, 64, 64, 64, 128, 1, 1, 0, False),        # ... two 32-channel chunks (the ring's minimum)
])
def test_conv_kernel(case):
    from ipdm_pytorch_amd import _lib
    # (the pointwise kernel takes a launch by its fill -- these shapes are small: pw_force drives them through it)
    with _lib.option("pw_force", 1 if CONV_CASE_KERNELS.get(case) == 10 else 0):
        if case in CONV_CASE_KERNELS:
            B, C1, C2, Hs, Ws, H, W, Cout, ks, stride = case[:10]
            got = _lib.lib().ipdm_conv_kernel_code(B, Cout, C1 + C2, ks, stride, H, W)
            assert got == CONV_CASE_KERNELS[case], (case, got)
        _conv_case(*case, seed=200 + sum(case[:8]))


def test_conv_kernel_random_shapes():
    """40 seeded random convolutions across every kernel family and edge condition at once: channel counts that are not
    multiples of the chunk / tile sizes, concat splits (aligned to the 8- or 32-channel chunk as the executor
    guarantees), ragged heights / widths (incl. W % 4 != 0: the dword epilogue), up-sampling to explicit sizes, stride 2,
    1x1, all three prologues, residual on / off."""
    rng = np.random.default_rng(20261003)
    chans = [1, 4, 8, 12, 16, 24, 36, 64, 72, 128, 136, 192, 256]
    for i in range(40):
        ks = int(rng.choice([3, 3, 3, 1]))
        stride = int(rng.choice([1, 1, 1, 2])) if ks == 3 else 1
        cout = int(rng.choice([1, 4, 8, 16, 24, 40, 64, 96, 128, 200, 256]))
        c1 = int(rng.choice(chans))
        c2 = int(rng.choice([0, 0, 8, 64, 128])) if c1 % 32 == 0 else 0
        act = int(rng.choice([0, 1, 2])) if (c1 + c2) > 1 else 0
        B = int(rng.integers(1, 4))
        Hs, Ws = int(rng.integers(5, 70)), int(rng.integers(5, 90))
        up = stride == 1 and rng.random() < 0.2
        H, W = (Hs * 2 - int(rng.integers(0, 2)), Ws * 2 - int(rng.integers(0, 2))) if up else (Hs, Ws)
        res = bool(rng.random() < 0.5)
        _conv_case(B, c1, c2, Hs, Ws, H, W, cout, ks, stride, act, res, seed=1000 + 17 * i)


@pytest.mark.parametrize("B,heads,T", [(1, 1, 35), (2, 4, 117), (1, 4, 1024), (1, 2, 1827), (1, 1, 7125)])
def test_attention_kernel(B, heads, T):
    from ipdm_pytorch_amd import _lib
    d = 64
    qkv = torch.from_numpy(synth.hash_normal((B, heads * 3 * d, T), 300 + T)) * 1.5
    out = torch.empty((B, heads * d, T), device=DEV)
    _lib.call("ipdm_op_attention", _lib.ptr(qkv.to(DEV)), _lib.ptr(out), B, heads, d, T, _lib.current_stream())
    q, k, v = qkv.reshape(B * heads, 3 * d, T).chunk(3, dim=1)
    scale = 1.0 / np.sqrt(np.sqrt(d))
    attn = torch.einsum("bct,bcs->bts", (q * scale).double(), (k * scale).double()).softmax(dim=-1)
    want = torch.einsum("bts,bcs->bct", attn, v.double()).reshape(B, heads * d, T).float()
    assert (out.cpu() - want).abs().max() <= 2e-5


@pytest.mark.parametrize("T", [1024, 1827, 4096, 640])
def test_attention_key_slices_do_not_depend_on_the_schedule(T):
    """Short sequences are reduced in key slices whose count depends on the layer alone; a batch that fills the chip walks
    the slices inside each workgroup, a single slice of the same layer runs them as separate workgroups + a combine pass.
    Both must give the same bits (sample i of the batch == sample i alone), and match torch."""
    from ipdm_pytorch_amd import _lib
    B, heads, d = 8, 4, 64
    qkv = (torch.from_numpy(synth.hash_normal((B, heads * 3 * d, T), 900 + T)) * 1.3).to(DEV)
    out = torch.empty((B, heads * d, T), device=DEV)
    _lib.call("ipdm_op_attention", _lib.ptr(qkv), _lib.ptr(out), B, heads, d, T, _lib.current_stream())
    for i in (0, 5):
        one = torch.empty((1, heads * d, T), device=DEV)
        qi = qkv[i:i + 1].contiguous()
        _lib.call("ipdm_op_attention", _lib.ptr(qi), _lib.ptr(one), 1, heads, d, T, _lib.current_stream())
        assert torch.equal(one[0], out[i]), (T, i, float((one[0] - out[i]).abs().max()))
    q, k, v = qkv[:1].cpu().reshape(heads, 3 * d, T).chunk(3, dim=1)
    scale = 1.0 / np.sqrt(np.sqrt(d))
    attn = torch.einsum("bct,bcs->bts", (q * scale).double(), (k * scale).double()).softmax(dim=-1)
    want = torch.einsum("bts,bcs->bct", attn, v.double()).reshape(1, heads * d, T).float()
    assert (out[:1].cpu() - want).abs().max() <= 2e-5


def test_attention_random_lengths():
    """Seeded random (B, heads, T): every ragged-tail length class of the 64-key tiles and 128/256-query workgroups."""
    from ipdm_pytorch_amd import _lib
    rng = np.random.default_rng(7)
    d = 64
    for T in [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511] + [int(v) for v in rng.integers(300, 2200, 6)]:
        B, heads = int(rng.integers(1, 4)), int(rng.integers(1, 5))
        qkv = torch.from_numpy(synth.hash_normal((B, heads * 3 * d, T), 900 + T)) * float(rng.uniform(0.3, 2.0))
        out = torch.full((B, heads * d, T), float("nan"), device=DEV)
        _lib.call("ipdm_op_attention", _lib.ptr(qkv.to(DEV)), _lib.ptr(out), B, heads, d, T, _lib.current_stream())
        q, k, v = qkv.reshape(B * heads, 3 * d, T).chunk(3, dim=1)
        scale = 1.0 / np.sqrt(np.sqrt(d))
        attn = torch.einsum("bct,bcs->bts", (q * scale).double(), (k * scale).double()).softmax(dim=-1)
        want = torch.einsum("bts,bcs->bct", attn, v.double()).reshape(B, heads * d, T).float()
        assert (out.cpu() - want).abs().max() <= 2e-5, (B, heads, T)


def test_attention_softmax_rescale_branch():
    """Online-softmax rescale: a late key block with a much larger score must take over (rule 26)."""
    from ipdm_pytorch_amd import _lib
    d, T = 64, 200
    qkv = torch.from_numpy(synth.hash_normal((1, 3 * d, T), 77)) * 0.3
    qkv[0, d:2 * d, 150] = qkv[0, 0:d, 10] * 40.0        # key 150 aligned with query 10 -> huge score late
    out = torch.empty((1, d, T), device=DEV)
    _lib.call("ipdm_op_attention", _lib.ptr(qkv.to(DEV)), _lib.ptr(out), 1, 1, d, T, _lib.current_stream())
    q, k, v = qkv.reshape(1, 3 * d, T).double().chunk(3, dim=1)
    attn = torch.einsum("bct,bcs->bts", q / 64 ** 0.25, k / 64 ** 0.25).softmax(dim=-1)
    want = torch.einsum("bts,bcs->bct", attn, v).float()
    assert (out.cpu() - want).abs().max() <= 2e-5
    assert (out.cpu()[0, :, 10] - v[0, :, 150].float()).abs().max() < 1e-3


def _op_conv(x, w, b, ks, act=0, gamma=None, beta=None, res=None, size=None, x2=None):
    """One fused convolution through the C ABI (ipdm_op_conv2d): [GN(+SiLU)] -> [nearest upsample] -> conv -> +bias [+res]."""
    from ipdm_pytorch_amd import _lib
    B, C1, Hs, Ws = x.shape
    C2 = 0 if x2 is None else x2.shape[1]
    H, W = size if size is not None else (Hs, Ws)
    Cout = w.shape[0]
    out = torch.empty((B, Cout, H, W), device=DEV)
    xd, x2d = x.to(DEV).contiguous(), (None if x2 is None else x2.to(DEV).contiguous())
    rd = None if res is None else res.to(DEV).contiguous()
    arrs = [None if t is None else np.ascontiguousarray(t.numpy(), dtype=np.float32) for t in (w, b, gamma, beta)]
    _lib.call("ipdm_op_conv2d", _lib.ptr(xd), C1, _lib.ptr(x2d), C2, B, Hs, Ws, H, W, _lib.ptr(arrs[0]), _lib.ptr(arrs[1]),
              Cout, ks, 1, act, ou.gn_groups(C1 + C2) if act else 0, _lib.ptr(arrs[2]), _lib.ptr(arrs[3]), _lib.ptr(rd),
              _lib.ptr(out), _lib.current_stream())
    return out


def test_reference_blocks_golden(golden):
    """The reference's own ResidualBlock (36 -> 24 channels, GroupNorm groups 36 / 24, 1x1 shortcut), Upsample (nearest to
    an explicit odd size + conv) and AttentionBlock (T = 35, 117) outputs (ops.npz, generated by importing the
    reference) against the same blocks composed from the library's fused kernels."""
    import torch.nn.functional as F
    from ipdm_pytorch_amd import _lib
    g = golden("ops")
    # ---- ResidualBlock (Model/model.py:95-130)
    keys = [str(k) for k in g["res_keys"]]
    shapes = dict(zip(keys, [(36,), (36,), (24, 36, 3, 3), (24,), (24, 64), (24,), (24,), (24,), (24, 24, 3, 3), (24,),
                             (24, 36, 1, 1), (24,)]))
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=25).items()}
    x = torch.from_numpy(synth.hash_normal((2, 36, 11, 9), 26))
    emb = torch.from_numpy(synth.hash_normal((1, 64), 27))
    bias1 = sd["conv1.2.bias"] + F.linear(F.silu(emb), sd["time_emb.1.weight"], sd["time_emb.1.bias"])[0]
    h1 = _op_conv(x, sd["conv1.2.weight"], bias1, 3, act=2, gamma=sd["conv1.0.weight"], beta=sd["conv1.0.bias"])
    sc = _op_conv(x, sd["shortcut.weight"], sd["shortcut.bias"], 1)
    y = _op_conv(h1.cpu(), sd["conv2.2.weight"], sd["conv2.2.bias"], 3, act=2, gamma=sd["conv2.0.weight"],
                 beta=sd["conv2.0.bias"], res=sc.cpu())
    np.testing.assert_allclose(y.cpu().numpy(), g["res_out"], rtol=0, atol=1e-5)
    # ---- Upsample (Model/model.py:160-171)
    up = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict({"conv.weight": (8, 8, 3, 3), "conv.bias": (8,)}, seed=23).items()}
    x = torch.from_numpy(synth.hash_normal((1, 8, 29, 63), 24))
    y = _op_conv(x, up["conv.weight"], up["conv.bias"], 3, size=(57, 125))
    np.testing.assert_allclose(y.cpu().numpy(), g["up_out"], rtol=0, atol=1e-5)
    # ---- AttentionBlock (Model/model.py:135-155)
    for tag, (C, heads, H, W) in {"attn64": (64, 1, 5, 7), "attn256": (256, 4, 9, 13)}.items():
        shapes = {"norm.weight": (C,), "norm.bias": (C,), "qkv.weight": (3 * C, C, 1, 1), "proj.weight": (C, C, 1, 1),
                  "proj.bias": (C,)}
        sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=21).items()}
        x = torch.from_numpy(synth.hash_normal((1, C, H, W), 22))
        qkv = _op_conv(x, sd["qkv.weight"], None, 1, act=1, gamma=sd["norm.weight"], beta=sd["norm.bias"])
        a = torch.empty((1, C, H * W), device=DEV)
        _lib.call("ipdm_op_attention", _lib.ptr(qkv.reshape(1, 3 * C, H * W).contiguous()), _lib.ptr(a), 1, heads,
                  C // heads, H * W, _lib.current_stream())
        y = _op_conv(a.reshape(1, C, H, W).cpu(), sd["proj.weight"], sd["proj.bias"], 1, res=x)
        np.testing.assert_allclose(y.cpu().numpy(), g[tag + "_out"], rtol=0, atol=1e-5)


STATS_CHAIN_CASES = [
    # B, C, H, W, CA, ksA, strideA, resA, act, CB        (conv A's kernel family decides the statistics geometry)
    (2, 64, 48, 64, 128, 3, 1, True, 2, 64),      # conv_ws 8x32x128 tiles, 16-byte epilogue, residual
    (2, 64, 40, 36, 64, 3, 1, False, 2, 64),      # conv_ws 16x32x64 tiles, ragged right edge and bottom (masked lanes)
    (1, 128, 21, 57, 256, 3, 1, True, 2, 40),     # width % 4 != 0: dword epilogue statistics
    (1, 96, 19, 33, 72, 3, 1, False, 1, 64),      # ragged couts (72 = 64 + 8): partial cout tile, GroupNorm groups 36
    (2, 64, 33, 47, 64, 3, 2, False, 2, 64),      # stride-2 producer (Downsample)
    (2, 256, 12, 20, 256, 1, 1, True, 2, 64),     # 1x1 producer with residual (attention proj)
    (8, 256, 32, 32, 256, 3, 1, True, 2, 64),     # the small-tile variant (4x32x64)
    (4, 128, 200, 96, 128, 3, 1, True, 2, 64),    # several rounds of tiles: statistics rows of a multi-round schedule
    (1, 256, 32, 32, 256, 3, 1, True, 2, 64),     # batch 1: K split, statistics from the combine pass
    (1, 256, 24, 40, 256, 1, 1, True, 2, 64),     # 1x1 producer, K split
    (2, 128, 26, 250, 128, 3, 1, True, 2, 64),    # width % 4 == 2 through the 16-byte epilogue: partial runs in the statistics
    (2, 128, 40, 125, 256, 3, 1, False, 2, 64),   # width % 4 == 1
    (2, 8, 70, 200, 8, 3, 1, True, 2, 8),         # direct narrow kernel 8 -> 8, ragged tiles
    (1, 16, 37, 130, 16, 3, 1, False, 2, 16),     # direct kernel 16 couts, width % 4 != 0
    (1, 1, 64, 72, 4, 3, 1, False, 2, 8),         # stem 1 -> 4
    (2, 16, 30, 44, 4, 1, 1, False, 1, 4),        # narrow 1x1
    (1, 48, 20, 24, 24, 3, 1, False, 2, 24),      # 16 < couts <= 32: the 4-wave kernel of conv.hip
    (2, 8, 37, 61, 8, 3, 2, False, 2, 8),         # narrow stride-2 (Downsample of the 4/8/16-channel levels): 4-wave kernel, ragged tiles
    (2, 256, 40, 72, 256, 1, 1, True, 2, 64),     # pointwise kernel (conv_pw) as the producer: a statistics row per 32 flat pixels, residual
    (3, 128, 37, 125, 128, 1, 1, False, 1, 64),   # ... ragged last item (4625 pixels), no residual
]


def _conv_gn_conv(case):
    """conv A -> GroupNorm(+SiLU) -> conv B where the GroupNorm statistics come from the per-tile partial sums conv A's
    epilogue wrote (ConvArgs::stats -> gn_tile_reduce -> gn_finalize), for every producing kernel family, against
    torch ops (fp32): 2e-5 relative like the single-convolution tests."""
    import ctypes
    import torch.nn.functional as F
    from ipdm_pytorch_amd import _lib
    B, C, H, W, CA, ksA, sA, resA, act, CB = case
    seed = 3000 + sum(case[:7])
    x = torch.from_numpy(synth.hash_normal((B, C, H, W), seed)) * 1.3 + 0.2
    wA = torch.from_numpy(synth.hash_normal((CA, C, ksA, ksA), seed + 1)) / np.sqrt(C * ksA * ksA)
    bA = torch.from_numpy(synth.hash_normal((CA,), seed + 2))
    wB = torch.from_numpy(synth.hash_normal((CB, CA, 3, 3), seed + 3)) / np.sqrt(CA * 9)
    bB = torch.from_numpy(synth.hash_normal((CB,), seed + 4))
    gamma = torch.from_numpy(synth.hash_uniform((CA,), seed + 5)) + 0.5
    beta = torch.from_numpy(synth.hash_normal((CA,), seed + 6)) * 0.2
    groups = ou.gn_groups(CA)
    mid = F.conv2d(x, wA, bA, stride=sA, padding=ksA // 2)
    r = torch.from_numpy(synth.hash_normal(tuple(mid.shape), seed + 7)) * 2 - 0.7 if resA else None
    if resA:
        mid = mid + r
    h = F.group_norm(mid, groups, gamma, beta, eps=1e-5)
    if act == 2:
        h = F.silu(h)
    want = F.conv2d(h, wB, bB, padding=1)
    d_mid = torch.full(tuple(mid.shape), float("nan"), device=DEV)
    d_out = torch.full(tuple(want.shape), float("nan"), device=DEV)
    xd = x.to(DEV)
    rd = r.to(DEV) if resA else None
    arrs = [np.ascontiguousarray(t.numpy(), dtype=np.float32) for t in (wA, bA, gamma, beta, wB, bB)]
    rows = ctypes.c_int32(-1)
    _lib.call("ipdm_op_conv_gn_conv", _lib.ptr(xd), C, B, H, W, _lib.ptr(arrs[0]), _lib.ptr(arrs[1]), CA, ksA, sA, _lib.ptr(rd),
              groups, _lib.ptr(arrs[2]), _lib.ptr(arrs[3]), act, _lib.ptr(arrs[4]), _lib.ptr(arrs[5]), CB, _lib.ptr(d_mid),
              _lib.ptr(d_out), ctypes.byref(rows), _lib.current_stream())
    assert rows.value > 0, rows.value                                  # every kernel family of the product path leaves fused statistics
    assert (d_mid.cpu() - mid).abs().max() <= 2e-5 * max(1.0, mid.abs().max().item())
    err = (d_out.cpu() - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), (err, case)
    return d_mid.cpu(), d_out.cpu(), rows.value


@pytest.mark.parametrize("case", STATS_CHAIN_CASES)
def test_fused_groupnorm_statistics_chain(case):
    _conv_gn_conv(case)


def test_conv_kernel_code_table():
    """ipdm_conv_kernel_code names the kernel a shape takes NOW (include/ipdm_hip.h): the production layers land where
    DESIGN section 3 says they do, so a parity case that believes it covers a kernel can assert it."""
    from ipdm_pytorch_amd import _lib
    code = _lib.lib().ipdm_conv_kernel_code
    table = [
        # B, Cout, Cin, ks, stride, H, W -> code
        ((8, 128, 128, 3, 1, 512, 512), 2),      # wide 3x3: Winograd, 128-cout tiles
        ((8, 256, 384, 3, 1, 114, 250), 2),
        ((1, 128, 128, 3, 1, 512, 512), 2),      # a lone slice still fills the chip at 512x512
        ((1, 256, 256, 3, 1, 64, 64), 1),        # ... not at 64x64: the 64-cout tiles (bit-identical results)
        ((8, 64, 64, 3, 1, 512, 512), 1),        # 64-cout layers: no 128-cout tile
        ((8, 128, 144, 3, 1, 228, 500), 2),      # 128 + 16 concat of the proj UNet
        ((8, 128, 72, 3, 1, 228, 500), 1),       # Cin not a multiple of 16
        ((8, 256, 256, 3, 1, 32, 32), 9),        # K-split layer: K slices inside conv_wino2 + combine pass
        ((8, 320, 256, 3, 1, 32, 32), 4),        # ... whose couts are off the 128-cout tile: the K-split direct kernel
        ((8, 768, 256, 1, 1, 64, 64), 10),       # qkv 1x1: the pointwise kernel
        ((8, 128, 256, 1, 1, 228, 500), 10),     # shortcut of the proj UNet's up path
        ((1, 768, 256, 1, 1, 57, 125), 3),       # ... a lone slice's launch would not fill the chip's waves: conv_ws (same bits)
        ((1, 128, 256, 1, 1, 228, 500), 10),     # ... this one does
        ((8, 768, 256, 1, 1, 16, 16), 4),        # low resolution: conv_ws with its K split
        ((8, 64, 128, 1, 1, 512, 512), 3),       # 64 couts: no whole 128-cout group
        ((8, 128, 128, 3, 2, 512, 512), 3),      # Downsample
        ((8, 8, 8, 3, 1, 2000, 912), 5),         # narrow level
        ((1, 24, 48, 3, 1, 20, 24), 8),          # 16 < Cout <= 32: the generic 4-wave kernel
    ]
    for args, want in table:
        assert code(*args) == want, (args, code(*args), want)
    with _lib.option("wino_v1", 1):
        assert code(8, 128, 128, 3, 1, 512, 512) == 1
    with _lib.option("conv_no_wino", 1):
        assert code(8, 128, 128, 3, 1, 512, 512) == 3
    with _lib.option("conv_no_pw", 1):
        assert code(8, 768, 256, 1, 1, 64, 64) == 3
    with _lib.option("pw_force", 1):
        assert code(1, 768, 256, 1, 1, 57, 125) == 10
    assert code(0, 128, 128, 3, 1, 8, 8) == -1
    # a layer that leaves fused statistics (its output feeds a GroupNorm) takes its kernel by a rule of the layer ALONE: the
    # plain query above answers by the batch's fill, the statistics query must not (the test aid mirrors the rule)
    scode = _lib.lib().ipdm_conv_kernel_code_stats
    assert code(8, 256, 256, 1, 1, 57, 125) == 10 and scode(8, 256, 256, 1, 1, 57, 125) == 3      # proj_out @57x125: 446 items per sample
    assert scode(1, 256, 256, 1, 1, 57, 125) == 3
    assert scode(1, 128, 256, 1, 1, 228, 500) == 10 and scode(8, 128, 256, 1, 1, 228, 500) == 10  # 3563 items per sample: either batch
    assert scode(8, 128, 128, 3, 1, 512, 512) == 2 and scode(8, 8, 8, 3, 1, 2000, 912) == 5        # (other families: as the plain query)


WINO128_CASES = [
    # B, C1, C2, H, W, Cout, act, res       (3x3 stride 1; whole 128-cout tiles, 16-channel chunks, more than 16 direct tiles
    #                                        per sample -- fewer are K-split layers, which stay on the direct kernel)
    (1, 128, 0, 72, 64, 128, 0, False),      # no prologue, no residual
    (2, 128, 0, 37, 145, 128, 2, True),      # ragged both ways, odd width (the partial-run epilogue)
    (1, 128, 128, 24, 70, 256, 2, False),    # concat, two cout tiles, width % 4 == 2
    (2, 64, 64, 45, 95, 128, 1, True),       # concat at a 64-channel boundary, width % 4 == 3
    (1, 256, 0, 17, 125, 256, 2, True),      # width % 4 == 1, 16 chunks
    (1, 32, 0, 33, 97, 128, 2, False),       # two chunks (the minimum)
    (3, 128, 16, 40, 104, 128, 2, True),     # concat with a 16-channel skip (proj UNet: 128 + 16 -> 128)
    (1, 128, 0, 130, 250, 128, 2, True),     # many tiles per workgroup, several rounds
]


@pytest.mark.parametrize("case", WINO128_CASES)
def test_wino_kernels_bit_identical(case):
    """The two Winograd kernels (round-3 64-cout tiles / round-4 128-cout tiles, conv_wino2.hip) accumulate every output in
    the same order, so the launcher may choose between them by the batch: their outputs must agree BIT FOR BIT, and both
    must sit within the usual 2e-5 of the fp32 torch op."""
    from ipdm_pytorch_amd import _lib
    import torch.nn.functional as F
    B, C1, C2, H, W, Cout, act, res = case
    seed = 8800 + sum(case[:6])
    Cin = C1 + C2
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed))
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)) * 2 + 0.5 if C2 else None
    w = torch.from_numpy(synth.hash_normal((Cout, Cin, 3, 3), seed + 2)) / np.sqrt(Cin * 9)
    bias = torch.from_numpy(synth.hash_normal((Cout,), seed + 3))
    gamma = torch.from_numpy(synth.hash_uniform((Cin,), seed + 4)) + 0.5
    beta = torch.from_numpy(synth.hash_normal((Cin,), seed + 5)) * 0.2
    groups = ou.gn_groups(Cin)
    h = x1 if x2 is None else torch.cat([x1, x2], 1)
    if act:
        h = F.group_norm(h, groups, gamma, beta, eps=1e-5)
        if act == 2:
            h = F.silu(h)
    want = F.conv2d(h, w, bias, padding=1)
    r = torch.from_numpy(synth.hash_normal(tuple(want.shape), seed + 6)) if res else None
    if res:
        want = want + r
    x1d, x2d, rd = x1.to(DEV), (x2.to(DEV) if C2 else None), (r.to(DEV) if res else None)
    wn, bn, gn_, ben = (np.ascontiguousarray(t.numpy()) for t in (w, bias, gamma, beta))
    outs = []
    for v1 in (1, 0):
        out = torch.full(tuple(want.shape), float("nan"), device=DEV)
        with _lib.option("wino_v1", v1), _lib.option("wino2_min_tiles", 1):
            code = _lib.lib().ipdm_conv_kernel_code(B, Cout, Cin, 3, 1, H, W)
            assert code == (1 if v1 else 2), code           # 1: conv_wino (64-cout tiles), 2: conv_wino2 (128-cout tiles)
            _lib.call("ipdm_op_conv2d", _lib.ptr(x1d), C1, _lib.ptr(x2d), C2, B, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1,
                      act, groups, _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max()
    err = (outs[1] - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), (err, case)


@pytest.mark.parametrize("case", WINO128_CASES + [
    (8, 128, 0, 64, 64, 256, 0, False),      # two cout tiles AND several rounds per workgroup: consecutive tiles of a workgroup switch weight tiles
    (8, 256, 0, 64, 64, 256, 2, True),       # (the first version took the new tile's weights for the last position of the old tile's last chunk)
])
def test_wino3_bf16x3_against_the_f32_kernels(case):
    """conv_wino3 (opt-in, option conv_bf16x3): conv_wino2's layers with the channel contraction on the bf16 matrix pipe through an
    error-free three-way split of both operands (six products, float32 accumulate).  Not the bits of the f32 kernels -- another
    summation order -- but the same function: within the 2e-5 of the fp32 torch op that every convolution kernel is held to, and
    as close to a float64 evaluation as conv_wino2 is (rms ratio <= 1.5 per case)."""
    from ipdm_pytorch_amd import _lib
    import torch.nn.functional as F
    B, C1, C2, H, W, Cout, act, res = case
    seed = 8800 + sum(case[:6])
    Cin = C1 + C2
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed))
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)) * 2 + 0.5 if C2 else None
    w = torch.from_numpy(synth.hash_normal((Cout, Cin, 3, 3), seed + 2)) / np.sqrt(Cin * 9)
    bias = torch.from_numpy(synth.hash_normal((Cout,), seed + 3))
    gamma = torch.from_numpy(synth.hash_uniform((Cin,), seed + 4)) + 0.5
    beta = torch.from_numpy(synth.hash_normal((Cin,), seed + 5)) * 0.2
    groups = ou.gn_groups(Cin)
    h = (x1 if x2 is None else torch.cat([x1, x2], 1)).double()
    if act:
        h = F.group_norm(h, groups, gamma.double(), beta.double(), eps=1e-5)
        if act == 2:
            h = F.silu(h)
    want = F.conv2d(h, w.double(), bias.double(), padding=1)
    r = torch.from_numpy(synth.hash_normal(tuple(want.shape), seed + 6)) if res else None
    if res:
        want = want + r.double()
    x1d, x2d, rd = x1.to(DEV), (x2.to(DEV) if C2 else None), (r.to(DEV) if res else None)
    wn, bn, gn_, ben = (np.ascontiguousarray(t.numpy()) for t in (w, bias, gamma, beta))
    outs = []
    for bf in (0, 1):
        out = torch.full(tuple(want.shape), float("nan"), device=DEV)
        with _lib.option("conv_bf16x3", bf), _lib.option("wino2_min_tiles", 1):
            code = _lib.lib().ipdm_conv_kernel_code(B, Cout, Cin, 3, 1, H, W)
            assert code == (12 if bf else 2), code          # 2: conv_wino2, 12: conv_wino3
            _lib.call("ipdm_op_conv2d", _lib.ptr(x1d), C1, _lib.ptr(x2d), C2, B, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1,
                      act, groups, _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
        outs.append(out.cpu().double())
    scale = max(1.0, want.abs().max().item())
    e2, e3 = (outs[0] - want), (outs[1] - want)
    assert e3.abs().max().item() <= 2e-5 * scale, (e3.abs().max().item(), e2.abs().max().item(), case)
    r2, r3 = e2.pow(2).mean().sqrt().item(), e3.pow(2).mean().sqrt().item()
    print("wino3 %s: |f64 - wino2| rms %.3e max %.3e | |f64 - wino3| rms %.3e max %.3e | ratio %.2f" % (case, r2, e2.abs().max().item(), r3, e3.abs().max().item(), r3 / r2))
    assert r3 <= 1.5 * r2, (r3, r2, case)


def test_wino128_fused_statistics_and_planar_reader():
    """conv_wino2 as the PRODUCER of fused GroupNorm statistics (conv A -> GroupNorm+SiLU -> conv B: d_mid, the rows and the
    result bit-equal to the 64-cout kernel's) and as the READER of a parity-planar x1 (the up2 -> concat -> conv chain)."""
    from ipdm_pytorch_amd import _lib
    # (more than 16 direct tiles per sample for BOTH convolutions: K-split layers take different kernels under the two switches)
    for case in [(2, 64, 80, 64, 128, 3, 1, True, 2, 128), (1, 128, 72, 57, 256, 3, 1, True, 2, 128), (2, 128, 26, 250, 128, 3, 1, True, 2, 128)]:
        res = []
        for v1 in (1, 0):
            with _lib.option("wino_v1", v1), _lib.option("wino2_min_tiles", 1):
                res.append(_conv_gn_conv(case))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and res[0][2] == res[1][2], case
    with _lib.option("wino2_min_tiles", 1):
        for case in UP2_CASES:
            if case[6] % 128 == 0 and case[7] == 3:
                test_upsample_conv_parity_form(case)


def _conv3x3_repeats(B, C1, C2, H, W, Cout, act, res, reps, seed=5):
    """One wide 3x3 layer `reps` + 1 times through conv_wino2: how many runs differ from the first; the first run; the 64-cout
    kernel's result (wino_v1) when the 128-cout kernel took the launch."""
    from ipdm_pytorch_amd import _lib
    Cin = C1 + C2
    x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed)).to(DEV)
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)).to(DEV) if C2 else None
    rd = torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)).to(DEV) if res else None
    wn, bn, gn_, ben = (np.ascontiguousarray(t, dtype=np.float32) for t in (
        synth.hash_normal((Cout, Cin, 3, 3), seed + 2) / np.sqrt(Cin * 9), synth.hash_normal((Cout,), seed + 3),
        synth.hash_uniform((Cin,), seed + 4) + 0.5, synth.hash_normal((Cin,), seed + 5) * 0.2))

    def once():
        out = torch.full((B, Cout, H, W), float("nan"), device=DEV)
        _lib.call("ipdm_op_conv2d", _lib.ptr(x1), C1, _lib.ptr(x2), C2, B, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1,
                  act, ou.gn_groups(Cin), _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
        return out
    with _lib.option("wino2_min_tiles", 1):
        code = _lib.lib().ipdm_conv_kernel_code(B, Cout, Cin, 3, 1, H, W)
        first = once()
        assert bool(torch.isfinite(first).all())
        bad = sum(int(not torch.equal(once(), first)) for _ in range(reps))
    ref = None
    if code == 2:
        with _lib.option("wino_v1", 1):
            ref = once()
    return bad, first, ref


def test_wino2_run_to_run_determinism():
    """conv_wino2 under repetition: launches with one tile per workgroup (waves end right behind their last stores), many-round
    launches, K slices, odd sizes -- every run bit-equal to the first and to the 64-cout kernel.  (The shelved 1x1 experiment
    on the same structure, tools/experiments/conv_pw.hip, failed exactly this.)"""
    for case in [(1, 128, 0, 72, 64, 128, 2, True), (2, 128, 0, 100, 96, 256, 2, False), (1, 256, 0, 32, 32, 256, 2, True),
                 (1, 128, 16, 61, 129, 128, 1, True)]:
        bad, first, ref = _conv3x3_repeats(*case, reps=12)
        assert bad == 0, case
        assert ref is None or torch.equal(ref, first), case


def test_wino3_fused_statistics_and_planar_reader():
    """conv_wino3 as the PRODUCER of fused GroupNorm statistics (conv A -> GroupNorm+SiLU -> conv B, both on conv_wino3 where
    eligible) and as the READER of a parity-planar x1 (the up2 -> concat -> conv chain): each against the torch ops, 2e-5."""
    from ipdm_pytorch_amd import _lib
    with _lib.option("conv_bf16x3", 1), _lib.option("wino2_min_tiles", 1):
        for case in [(2, 64, 80, 64, 128, 3, 1, True, 2, 128), (1, 128, 72, 57, 256, 3, 1, True, 2, 128), (2, 128, 26, 250, 128, 3, 1, True, 2, 128),
                     (8, 128, 64, 64, 128, 3, 1, True, 2, 256)]:
            _conv_gn_conv(case)
        for case in UP2_CASES:
            if case[6] % 128 == 0 and case[7] == 3:
                _up_conv_chain(case)


def test_wino3_run_to_run_determinism():
    """conv_wino3 (option conv_bf16x3) under repetition -- the kind of test that found its first version wrong: with conv_wino2's load
    order one output row of a tile came out as garbage in ~1 of 100 tiles, run-dependently (cause unidentified; the shipped order and
    the pinned operand registers are an empirical fix: NOTEBOOK.md round 6).  Shapes with one tile per workgroup and with many rounds,
    ragged edges, concat, residual; sixty launches each, every run bit-equal to the first, and the first within 2e-5 of conv_wino2."""
    from ipdm_pytorch_amd import _lib
    for case in [(2, 128, 0, 37, 145, 128, 2, True), (1, 128, 0, 130, 250, 128, 2, True), (2, 128, 0, 36, 144, 128, 0, False),
                 (2, 64, 64, 45, 95, 128, 1, True), (8, 128, 0, 128, 128, 128, 2, True), (3, 128, 16, 40, 104, 128, 2, True)]:
        with _lib.option("conv_bf16x3", 1):
            assert _lib.lib().ipdm_conv_kernel_code(case[0], case[5], case[1] + case[2], 3, 1, case[3], case[4]) == 12, case
            bad, first, _ = _conv3x3_repeats(*case, reps=60)
        _, ref, _ = _conv3x3_repeats(*case, reps=0)
        assert bad == 0, (bad, case)
        assert (first - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), case


def test_wino3_batch_is_its_slices():
    """Under option conv_bf16x3 the kernel choice is a rule of the layer ALONE (conv_wino3's bits are not the float32 Winograd
    kernels', so -- unlike the choice between those two -- it must not look at the batch): a batch's result is, bit for bit, its
    slices' results run alone, also on the levels where the default rule gives a lone slice the 64-cout kernel (64 tiles per
    sample at 256 -> 256 @64x64).  The first version of the option broke this: the whole suite run under IPDM_CONV_BF16X3=1
    (profiles/r06j_suite_bf16x3.txt) found a B = 2 batch 2.9e-6 away from its slices."""
    from ipdm_pytorch_amd import _lib
    code = _lib.lib().ipdm_conv_kernel_code
    for case in [(8, 256, 0, 64, 64, 256, 2, True), (8, 128, 16, 40, 104, 128, 2, True), (3, 128, 128, 125, 57, 128, 1, False)]:
        B, C1, C2, H, W, Cout, act, res = case
        Cin, seed = C1 + C2, 8900 + sum(case[:6])
        assert code(1, Cout, Cin, 3, 1, H, W) == 1 and code(8, Cout, Cin, 3, 1, H, W) == 2, case     # (the default rule looks at the batch)
        x1 = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed)).to(DEV)
        x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)).to(DEV) if C2 else None
        rd = torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)).to(DEV) if res else None
        wn, bn, gn_, ben = (np.ascontiguousarray(t, dtype=np.float32) for t in (
            synth.hash_normal((Cout, Cin, 3, 3), seed + 2) / np.sqrt(Cin * 9), synth.hash_normal((Cout,), seed + 3),
            synth.hash_uniform((Cin,), seed + 4) + 0.5, synth.hash_normal((Cin,), seed + 5) * 0.2))

        def run(lo, hi):
            out = torch.full((hi - lo, Cout, H, W), float("nan"), device=DEV)
            a, b, r = x1[lo:hi].contiguous(), (x2[lo:hi].contiguous() if C2 else None), (rd[lo:hi].contiguous() if res else None)
            _lib.call("ipdm_op_conv2d", _lib.ptr(a), C1, _lib.ptr(b), C2, hi - lo, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, 3, 1,
                      act, ou.gn_groups(Cin), _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(r), _lib.ptr(out), _lib.current_stream())
            return out
        with _lib.option("conv_bf16x3", 1):
            assert code(1, Cout, Cin, 3, 1, H, W) == 12 and code(B, Cout, Cin, 3, 1, H, W) == 12, case
            whole = run(0, B)
            for i in range(B):
                assert torch.equal(run(i, i + 1)[0], whole[i]), (case, i)
        ref = run(0, B)                                                                                 # (conv_wino2)
        assert not torch.equal(ref, whole) and (ref - whole).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), case


PW_CASES = [
    # B, C1, C2, H, W, Cout, act, res        (1x1; whole 128-cout groups, 32-channel chunks, not a K-split layer)
    (2, 256, 0, 57, 125, 768, 1, False),     # qkv at T = 7125: six cout tiles per pixel tile, GroupNorm table, ragged last item
    (2, 256, 0, 57, 125, 256, 0, True),      # proj_out + residual
    (2, 128, 128, 45, 95, 128, 0, False),    # concat
    (3, 96, 32, 41, 67, 128, 1, True),       # concat at a 96-channel boundary, GroupNorm over both sources, three samples
    (1, 128, 0, 512, 512, 128, 0, True),     # several items per wave
    (8, 256, 0, 64, 64, 768, 1, False),      # the img UNet's qkv at batch 8: the table is rewritten when a wave's sample changes
]


@pytest.mark.parametrize("case", PW_CASES)
def test_pointwise_kernel_bit_identical_to_the_staged_one(case):
    """conv_pw.hip (a wave per 32 pixels x 128 couts, operands straight from memory, no barrier) accumulates every output
    in the order conv_ws.hip's 1x1 path does -- channels ascending as an exact fmaf chain, + bias, + residual -- so the two
    agree bit for bit, run after run; and both agree with torch (fp32) to 2e-5 relative."""
    import torch.nn.functional as F
    from ipdm_pytorch_amd import _lib
    B, C1, C2, H, W, Cout, act, res = case
    seed = 9100 + sum(case[:6])
    x = torch.from_numpy(synth.hash_normal((B, C1, H, W), seed)) * 1.4 + 0.3
    x2 = torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)) * 0.8 if C2 else None
    w = torch.from_numpy(synth.hash_normal((Cout, C1 + C2, 1, 1), seed + 2)) / np.sqrt(C1 + C2)
    b = torch.from_numpy(synth.hash_normal((Cout,), seed + 3))
    gamma = torch.from_numpy(synth.hash_uniform((C1 + C2,), seed + 4)) + 0.5 if act else None
    beta = torch.from_numpy(synth.hash_normal((C1 + C2,), seed + 5)) * 0.2 if act else None
    r = torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)) if res else None
    code = _lib.lib().ipdm_conv_kernel_code
    with _lib.option("pw_force", 1):     # (the test shapes are small: by its fill rule the kernel would leave some of them to conv_ws)
        assert code(B, Cout, C1 + C2, 1, 1, H, W) == 10
        got = _op_conv(x, w, b, 1, act=act, gamma=gamma, beta=beta, res=r, x2=x2)
        for rep in range(4):             # run after run, and with either item shape (32 / 64 pixels per wave)
            with _lib.option("pw_item", 1 + rep % 2):
                assert torch.equal(_op_conv(x, w, b, 1, act=act, gamma=gamma, beta=beta, res=r, x2=x2), got)
    with _lib.option("conv_no_pw", 1):
        assert code(B, Cout, C1 + C2, 1, 1, H, W) == 3
        staged = _op_conv(x, w, b, 1, act=act, gamma=gamma, beta=beta, res=r, x2=x2)
    assert torch.equal(got, staged)
    h = x if x2 is None else torch.cat([x, x2], 1)
    if act:
        h = F.group_norm(h, ou.gn_groups(C1 + C2), gamma, beta, eps=1e-5)
    want = F.conv2d(h, w, b)
    if res:
        want = want + r
    assert (got.cpu() - want).abs().max() <= 2e-5 * max(1.0, want.abs().max().item())


def test_pointwise_kernel_statistics_and_planar_reader_equal_the_staged_kernel():
    """The two chains the executor builds around a 1x1 layer -- producer of fused GroupNorm statistics, reader of a
    parity-planar Upsample output -- give the same bits with either kernel behind the 1x1 layer."""
    from ipdm_pytorch_amd import _lib
    _lib.set_option("pw_force", 1)
    try:
        _pw_chains()
    finally:
        _lib.set_option("pw_force", 0)


def _pw_inputs(case):
    B, C1, C2, H, W, Cout, act, res = case
    seed = 9100 + sum(case[:6])
    d = dict(x=(torch.from_numpy(synth.hash_normal((B, C1, H, W), seed)) * 1.4 + 0.3).to(DEV),
             x2=(torch.from_numpy(synth.hash_normal((B, C2, H, W), seed + 1)) * 0.8).to(DEV) if C2 else None,
             w=np.ascontiguousarray(synth.hash_normal((Cout, C1 + C2, 1, 1), seed + 2) / np.sqrt(C1 + C2), dtype=np.float32),
             b=np.ascontiguousarray(synth.hash_normal((Cout,), seed + 3), dtype=np.float32),
             gamma=np.ascontiguousarray(synth.hash_uniform((C1 + C2,), seed + 4) + 0.5, dtype=np.float32) if act else None,
             beta=np.ascontiguousarray(synth.hash_normal((C1 + C2,), seed + 5) * 0.2, dtype=np.float32) if act else None,
             r=torch.from_numpy(synth.hash_normal((B, Cout, H, W), seed + 6)).to(DEV) if res else None)
    return d


def _pw_run(handle, case, d):
    """ipdm_op_conv2d of a 1x1 layer through the C ABI of `handle` (the product library or a variant build of it)."""
    from ipdm_pytorch_amd import _lib
    B, C1, C2, H, W, Cout, act, res = case
    out = torch.full((B, Cout, H, W), float("nan"), device=DEV)
    rc = handle.ipdm_op_conv2d(_lib.ptr(d["x"]), C1, _lib.ptr(d["x2"]), C2, B, H, W, H, W, _lib.ptr(d["w"]), _lib.ptr(d["b"]), Cout, 1, 1,
                               act, ou.gn_groups(C1 + C2) if act else 0, _lib.ptr(d["gamma"]), _lib.ptr(d["beta"]), _lib.ptr(d["r"]),
                               _lib.ptr(out), _lib.current_stream())
    assert rc == 0, handle.ipdm_last_error()
    return out


# the table is rewritten when a wave's sample changes (three samples, 12 items per sample and cout tile: waves cross samples
# mid-stream); a ragged last item (41 x 67 = 2747 pixels); several items per wave; both sources of a concat
PW_STRESS_CASES = [(3, 96, 32, 41, 67, 128, 1, False), (2, 256, 0, 57, 125, 768, 1, False), (2, 256, 0, 57, 125, 256, 0, True),
                   (1, 128, 128, 96, 96, 128, 0, True)]


def test_pointwise_kernel_under_repetition():
    """the shipped pointwise kernel keeps both MFMA operands in a register ring that inline asm loads and
    hand-counted s_waitcnt vmcnt(N) wait for -- an idiom whose shelved first version (tools/experiments/conv_pw.hip) produced a
    run-dependent handful of wrong zeros.  200 launches per case and item shape (GroupNorm table rewritten mid-wave, ragged
    last item, residual, concat): every output equals the first bit for bit, and the first equals the staged kernel's."""
    from ipdm_pytorch_amd import _lib
    h = _lib.lib()
    for case in PW_STRESS_CASES:
        d = _pw_inputs(case)
        with _lib.option("conv_no_pw", 1):
            staged = _pw_run(h, case, d)
        assert bool(torch.isfinite(staged).all())
        with _lib.option("pw_force", 1):
            assert h.ipdm_conv_kernel_code(case[0], case[5], case[1] + case[2], 1, 1, case[3], case[4]) == 10
            for item in (1, 2):
                with _lib.option("pw_item", item):
                    bad = sum(int(not torch.equal(_pw_run(h, case, d), staged)) for _ in range(100))
                assert bad == 0, (case, item, bad)


def test_pointwise_ring_waits_equal_a_full_drain():
    """... and against a build of the SAME kernel in which every ring wait drains the whole queue (libipdm_hip_pwsafe.so:
    conv_pw.hip with -DIPDM_PW_SAFE_WAIT, `make pwsafe`, built by __graft_entry__.build()): a hand-counted wait that is one
    too loose shows as a difference between the two builds; a toolchain bump cannot pass silently."""
    import ctypes as C
    from ipdm_pytorch_amd import _lib
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libipdm_hip_pwsafe.so")
    assert os.path.isfile(path), "libipdm_hip_pwsafe.so missing: run __graft_entry__.build()"
    safe = C.CDLL(path)
    for name in ("ipdm_op_conv2d", "ipdm_set_option", "ipdm_conv_kernel_code"):
        getattr(safe, name).restype, getattr(safe, name).argtypes = _lib.PROTOTYPES[name]
    safe.ipdm_last_error.restype = C.c_char_p
    assert safe.ipdm_set_option(b"pw_force", 1) == 0
    h = _lib.lib()
    with _lib.option("pw_force", 1):
        for case in PW_STRESS_CASES + [PW_CASES[4], PW_CASES[5]]:
            d = _pw_inputs(case)
            assert safe.ipdm_conv_kernel_code(case[0], case[5], case[1] + case[2], 1, 1, case[3], case[4]) == 10
            for item in (1, 2):
                assert safe.ipdm_set_option(b"pw_item", item) == 0
                want = _pw_run(safe, case, d)
                with _lib.option("pw_item", item):
                    for rep in range(10):
                        got = _pw_run(h, case, d)
                        assert torch.equal(got, want), (case, item, rep, int((got != want).sum()))


def _pw_chains():
    from ipdm_pytorch_amd import _lib
    for case in STATS_CHAIN_CASES[-2:]:
        with _lib.option("pw_item", 1):
            mid1, out1, _ = _conv_gn_conv(case)
        with _lib.option("pw_item", 2):
            mid, out, rows = _conv_gn_conv(case)
        assert torch.equal(mid, mid1) and torch.equal(out, out1)      # the statistics rows do not depend on the item shape
        assert rows == -(-case[2] * case[3] // 32)          # a row per 32 flat pixels
        with _lib.option("conv_no_pw", 1):
            mid_s, out_s, rows_s = _conv_gn_conv(case)
        assert rows_s == case[2] * -(-case[3] // 32)         # (conv_ws.hip: a row per pixel row and 32-pixel tile column)
        assert torch.equal(mid, mid_s)
        # (the statistics are sums over different pixel sets: the normalised result agrees to rounding, not bit for bit)
        assert (out - out_s).abs().max() <= 2e-5 * max(1.0, out_s.abs().max().item())
    for case in UP2_CASES[-2:]:
        with _lib.option("conv_no_pw", 1):
            mid_s, out_s = _up_conv_chain(case)
        for item in (1, 2):
            with _lib.option("pw_item", item):
                mid, out = _up_conv_chain(case)
            assert torch.equal(mid, mid_s) and torch.equal(out, out_s)


def test_profile_classes_mask():
    """ipdm_profile_begin_classes records only the classes asked for (bench.py times its headline with the dominant kernel's
    classes and the rest on an extra step): a pointwise launch is class 1, a wide 3x3 launch class 5 (Winograd, 128-cout
    tiles); a too-short result array is refused."""
    import ctypes as C
    from ipdm_pytorch_amd import _lib
    x = torch.from_numpy(synth.hash_normal((1, 128, 64, 64), 77))
    w1 = torch.from_numpy(synth.hash_normal((128, 128, 1, 1), 78)) / 12
    w3 = torch.from_numpy(synth.hash_normal((128, 128, 3, 3), 79)) / 34
    b = torch.zeros(128)
    NC = _lib.PROF_CLASSES

    def run(mask):
        fl, ms, nl = (C.c_double * NC)(), (C.c_double * NC)(), (C.c_int64 * NC)()
        if mask is None:
            _lib.call("ipdm_profile_begin", 64)
        else:
            _lib.call("ipdm_profile_begin_classes", 64, mask)
        with _lib.option("wino2_min_tiles", 1):
            _op_conv(x, w1, b, 1)
            _op_conv(x, w3, b, 3)
        torch.cuda.synchronize()
        _lib.call("ipdm_profile_end", C.byref(fl), C.byref(ms), C.byref(nl), NC)
        return list(nl), list(fl), list(ms)

    nl, fl, ms = run(None)
    assert nl[1] == 1 and nl[5] == 1 and sum(nl) == 2
    assert fl[1] == 2.0 * 64 * 64 * 128 * 128 and fl[5] == 2.0 * 32 * 32 * 16 * 128 * 128 and ms[1] > 0 and ms[5] > 0
    nl, _, _ = run(1 << 1)
    assert nl[1] == 1 and sum(nl) == 1
    nl, _, _ = run((1 << 5) | (1 << 3) | (1 << 0))
    assert nl[5] == 1 and sum(nl) == 1
    # a wide Upsample layer is class 7 (conv_wup2: executed flops = 9 products per source pixel), class 1 on the 2x2-tap kernel
    for off, cls, taps in ((0, 7, 9), (1, 1, 16)):
        with _lib.option("conv_no_wup2", off):
            _lib.call("ipdm_profile_begin", 64)
            _up_conv_chain((2, 128, 16, 32, 128, 0, 128, 3, 2))
            torch.cuda.synchronize()
            fl, ms, nl = (C.c_double * NC)(), (C.c_double * NC)(), (C.c_int64 * NC)()
            _lib.call("ipdm_profile_end", C.byref(fl), C.byref(ms), C.byref(nl), NC)
        assert nl[cls] == 1 and fl[cls] == 2.0 * 2 * 16 * 32 * 128 * 128 * taps, (off, list(nl), list(fl))
    fl, ms, nl = (C.c_double * NC)(), (C.c_double * NC)(), (C.c_int64 * NC)()
    _lib.call("ipdm_profile_begin", 8)
    with pytest.raises(RuntimeError):          # the caller states its array length: one shorter than the class count is refused
        _lib.call("ipdm_profile_end", C.byref(fl), C.byref(ms), C.byref(nl), NC - 1)
    _lib.call("ipdm_profile_end", C.byref(fl), C.byref(ms), C.byref(nl), NC)


def test_direct_fallback_of_the_winograd_layers():
    """conv_no_wino (per call; bench.py's '-nowino' mode): the layers the Winograd kernels take by default on the direct
    implicit-GEMM kernel (conv_ws) -- plain, concat, fused-statistics producer, parity-planar reader -- so that the shipped
    fallback stays covered now that every wide op-level case uploads Winograd weights."""
    from ipdm_pytorch_amd import _lib
    with _lib.option("conv_no_wino", 1):
        assert _lib.lib().ipdm_conv_kernel_code(8, 128, 128, 3, 1, 64, 96) == 3
        for i, case in enumerate([(8, 128, 0, 64, 96, 64, 96, 128, 3, 1, 2, True), (2, 64, 64, 48, 40, 48, 40, 64, 3, 1, 2, True),
                                  (2, 128, 0, 19, 250, 19, 250, 128, 3, 1, 2, True), (2, 128, 64, 203, 90, 203, 90, 256, 3, 1, 2, False)]):
            _conv_case(*case, seed=7300 + i)
        _conv_gn_conv((4, 128, 200, 96, 128, 3, 1, True, 2, 64))
        _conv_gn_conv((2, 128, 26, 250, 128, 3, 1, True, 2, 64))
        test_upsample_conv_parity_form((1, 128, 40, 72, 128, 64, 128, 3, 2))
        test_upsample_conv_parity_form((2, 64, 33, 47, 64, 0, 64, 3, 2))


def test_k_split_layers_on_the_winograd_kernel():
    """Layers the direct tiling splits along K (<= 16 tiles of 8x32x128 per sample) run in the Winograd domain with the K
    slices INSIDE conv_wino2 (round 4; a rule of the layer alone, so batch 1 and batch 8 take the same slices): same
    tolerance, same combine pass (bias, residual, statistics rows) as the K-split direct kernel, which stays the path of the
    layers conv_wino2 cannot slice (Cout or Cin off its tile sizes) and of the wino_v1 / conv_no_wino arms."""
    from ipdm_pytorch_amd import _lib
    code = _lib.lib().ipdm_conv_kernel_code
    assert code(1, 256, 256, 3, 1, 32, 32) == 9 and code(8, 256, 256, 3, 1, 32, 32) == 9 and code(8, 256, 512, 3, 1, 63, 29) == 9
    assert code(1, 192, 256, 3, 1, 32, 32) == 4                      # 192 couts: no whole 128-cout tiles -> K-split direct kernel
    with _lib.option("conv_no_wino", 1):
        assert code(8, 256, 256, 3, 1, 32, 32) == 4
    for i, case in enumerate([(1, 256, 0, 32, 32, 32, 32, 256, 3, 1, 2, True), (1, 256, 256, 16, 24, 16, 24, 256, 3, 1, 2, False),
                              (2, 256, 0, 63, 29, 63, 29, 256, 3, 1, 2, True), (8, 256, 128, 29, 63, 29, 63, 256, 3, 1, 1, True),
                              (1, 128, 0, 20, 36, 20, 36, 128, 3, 1, 0, False)]):
        _conv_case(*case, seed=7100 + i)
    for case in [(1, 256, 32, 32, 256, 3, 1, True, 2, 64), (2, 256, 29, 63, 256, 3, 1, True, 2, 128)]:
        d_mid, d_out, rows = _conv_gn_conv(case)
        assert rows == -(-case[2] * case[3] // 2048), rows                # the combine pass's statistics rows (SPLIT_PIX pixels each)
    # a batch is its slices: the K split does not look at the batch
    one = _conv_case_out((1, 256, 0, 32, 32, 32, 32, 256, 3, 1, 2, True), 7100)
    four = _conv_case_out((4, 256, 0, 32, 32, 32, 32, 256, 3, 1, 2, True), 7100)
    assert torch.equal(one[0], four[0])


def test_narrow_convolutions_on_the_16_cout_mfma_opt_in():
    """conv_nm.hip (option conv_nm, off by default: NOTEBOOK.md, negative results): the narrow stride-1 layers on
    v_mfma_f32_16x16x4_f32 -- single convolutions (3x3 / 1x1, 8 and 16 couts, concat, every channel-group count, ragged
    strips, widths that are not multiples of 4, all prologues, residual) and the fused-statistics chain -- against the
    same torch references and tolerance as the default kernels."""
    from ipdm_pytorch_amd import _lib
    with _lib.option("conv_nm", 2):
        for i, case in enumerate([
                # B, C1, C2, Hs, Ws, H, W, Cout, ks, stride, act, res
                (1, 8, 0, 16, 64, 16, 64, 8, 3, 1, 0, False),
                (2, 16, 0, 37, 45, 37, 45, 16, 3, 1, 2, True),        # ragged both ways, element-wise stores
                (1, 16, 0, 50, 200, 50, 200, 16, 3, 1, 2, True),      # several strips, W % 64 != 0
                (2, 8, 4, 33, 132, 33, 132, 8, 3, 1, 2, False),       # concat 8 + 4 (3 groups)
                (1, 16, 16, 70, 128, 70, 128, 16, 3, 1, 2, False),    # 8 groups
                (1, 16, 8, 64, 192, 64, 192, 8, 3, 1, 1, False),      # 6 groups, GroupNorm without SiLU
                (1, 4, 0, 40, 72, 40, 72, 8, 3, 1, 2, False),         # one group
                (1, 8, 0, 20, 68, 20, 68, 16, 3, 1, 2, False),
                (2, 16, 8, 35, 140, 35, 140, 8, 1, 1, 0, False),      # 1x1 shortcuts
                (1, 16, 0, 47, 61, 47, 61, 16, 1, 1, 0, True)]):
            _conv_case(*case, seed=7000 + i)
        for case in [(2, 16, 70, 200, 16, 3, 1, True, 2, 16), (1, 16, 37, 130, 16, 3, 1, False, 2, 16),
                     (2, 8, 70, 200, 8, 3, 1, True, 2, 8), (1, 8, 33, 61, 16, 1, 1, False, 1, 16)]:
            test_fused_groupnorm_statistics_chain(case)


@pytest.mark.parametrize("offset", [30.0, 300.0])
def test_fused_groupnorm_statistics_with_a_large_channel_offset(offset):
    """Fused statistics are float32 per-tile {sum, sum of squares} folded in float64 (var = E[v^2] - mean^2).  With a
    channel mean far above its spread -- here conv A's bias puts |mean|/std at ~30 and ~300, far beyond what a GroupNorm
    input of this network family shows -- the cancellation costs relative accuracy of the variance ~1e-9 (mean/std)^2
    (per-tile rounding averaged over thousands of tiles), i.e. a scale error of the normalised values that must stay below
    the representation noise of the float32 activations themselves, 6e-8 |mean|/std per element.  Reference: float64."""
    import ctypes
    import torch.nn.functional as F
    from ipdm_pytorch_amd import _lib
    B, C, H, W, CA, CB = 2, 64, 64, 96, 64, 64
    x = torch.from_numpy(synth.hash_normal((B, C, H, W), 7001))
    wA = torch.from_numpy(synth.hash_normal((CA, C, 3, 3), 7002)) / np.sqrt(C * 9)
    bA = offset * (1 + 0.1 * torch.from_numpy(synth.hash_normal((CA,), 7003)))
    wB = torch.from_numpy(synth.hash_normal((CB, CA, 3, 3), 7004)) / np.sqrt(CA * 9)
    bB = torch.from_numpy(synth.hash_normal((CB,), 7005))
    gamma = torch.from_numpy(synth.hash_uniform((CA,), 7006)) + 0.5
    beta = torch.from_numpy(synth.hash_normal((CA,), 7007)) * 0.2
    groups = ou.gn_groups(CA)
    d_mid = torch.full((B, CA, H, W), float("nan"), device=DEV)
    d_out = torch.full((B, CB, H, W), float("nan"), device=DEV)
    arrs = [np.ascontiguousarray(t.numpy(), dtype=np.float32) for t in (wA, bA, gamma, beta, wB, bB)]
    rows = ctypes.c_int32(-1)
    xd = x.to(DEV)
    _lib.call("ipdm_op_conv_gn_conv", _lib.ptr(xd), C, B, H, W, _lib.ptr(arrs[0]), _lib.ptr(arrs[1]), CA, 3, 1, None,
              groups, _lib.ptr(arrs[2]), _lib.ptr(arrs[3]), 2, _lib.ptr(arrs[4]), _lib.ptr(arrs[5]), CB, _lib.ptr(d_mid),
              _lib.ptr(d_out), ctypes.byref(rows), _lib.current_stream())
    assert rows.value > 0
    # float64 reference computed FROM THE DEVICE'S OWN float32 mid (so only GroupNorm + conv B are under test)
    mid = d_mid.cpu().double()
    h = F.silu(F.group_norm(mid, groups, gamma.double(), beta.double(), eps=1e-5))
    want = F.conv2d(h, wB.double(), bB.double(), padding=1)
    ratio = float((mid.mean(dim=(2, 3)).abs() / mid.std(dim=(2, 3))).max())     # per-channel |mean|/std; groups of 2 channels
    err = float((d_out.cpu().double() - want).abs().max())
    budget = (2e-5 + 2e-9 * ratio * ratio) * max(1.0, float(want.abs().max()))
    print("offset %g: |mean|/std %.0f, err %.3e, budget %.3e" % (offset, ratio, err, budget))
    assert err <= budget, (err, budget, ratio)


UP2_CASES = [
    # B, C, Hs, Ws, CA, C2, CB, ksB, act
    (2, 128, 16, 32, 128, 0, 128, 3, 2),          # 128-cout tiles; GroupNorm from the parity form's fused statistics
    (1, 64, 13, 21, 64, 64, 64, 3, 2),            # 64-cout tiles, odd source size (ragged tiles in every parity), concat with a skip
    (2, 256, 9, 17, 256, 128, 128, 1, 2),         # 1x1 reader (the shortcut of the next block) over cat(mid, skip)
    (1, 128, 12, 20, 128, 16, 16, 3, 2),          # narrow reader (144 -> 16: the direct kernel) of a parity-planar source
    (1, 128, 12, 20, 128, 16, 16, 1, 0),          # its 1x1 shortcut
    (1, 16, 10, 12, 16, 0, 16, 3, 2),             # narrow Upsample: the direct kernel's parity form (NCHW output)
    (2, 16, 37, 45, 16, 8, 8, 3, 2),              # the same with ragged tiles on both axes (74x90 outputs) and a concatenated skip
    (1, 8, 20, 70, 8, 0, 8, 1, 1),                # 8 channels (the CO = 8 instantiation), 1x1 reader
    (1, 4, 9, 11, 4, 0, 4, 3, 0),                 # 4 channels: not eligible, 3x3 form with nearest addressing
    (1, 256, 4, 4, 256, 0, 256, 3, 1),            # 256 channels at 8x8: reader with a K split
    (1, 256, 8, 8, 256, 256, 256, 3, 2),          # ... K slices inside conv_wino2 over cat(parity-planar, NCHW skip): slices 2, 3 begin in the skip half
    (1, 128, 228, 500, 128, 16, 16, 3, 2),        # production size (transposed sinogram level 500x228 -> 1000x456) and its 144 -> 16 reader
    (1, 128, 40, 72, 128, 64, 128, 3, 2),         # Winograd-domain reader of a parity-planar source, concatenated skip (80x144)
    (2, 64, 33, 47, 64, 0, 64, 3, 2),             # ... ragged tiles on both axes (66x94), border tiles on all sides
    (1, 64, 34, 50, 64, 64, 64, 3, 1),            # ... GroupNorm without SiLU, 68x100
    (2, 128, 24, 40, 128, 128, 128, 1, 1),        # pointwise kernel reading cat(parity-planar, NCHW skip) with GroupNorm (48x80)
    (1, 128, 27, 45, 128, 0, 256, 1, 0),          # ... a parity-planar source alone, ragged last item (54x90 = 4860 pixels)
    # the F(2x2,2x2) form's ragged right edge with 2 and 3 pixels of a 4-pixel run inside (1: the 45-column cases above), fused
    # statistics over several tile columns and clipped tile rows; two cout tiles
    (1, 128, 9, 34, 128, 0, 128, 3, 2),
    (2, 128, 7, 39, 128, 0, 128, 3, 2),
    (1, 256, 10, 70, 256, 0, 128, 3, 1),
    (1, 128, 114, 250, 128, 0, 128, 1, 2),        # production size (sinogram level 250x114 -> 500x228), 2 pixels of the last run inside
]


def _up_conv_chain(case):
    """Upsample (nearest 2x + 3x3 conv) evaluated as four 2x2-tap convolutions over the source grid (the taps that fall on
    one source pixel added up when the weights are packed; on narrow levels inside the direct kernel), its parity-planar output, the fused statistics of that output
    and every kind of reader (3x3 / 1x1 wave-specialised kernels with and without a concatenated skip, the narrow direct
    kernel, a K-split layer) against torch ops in float32: 2e-5 relative like the other convolution tests."""
    import ctypes
    import torch.nn.functional as F
    from ipdm_pytorch_amd import _lib
    B, C, Hs, Ws, CA, C2, CB, ksB, act = case
    seed = 5000 + sum(case)
    x = torch.from_numpy(synth.hash_normal((B, C, Hs, Ws), seed)) * 1.1 + 0.3
    wA = torch.from_numpy(synth.hash_normal((CA, C, 3, 3), seed + 1)) / np.sqrt(C * 9)
    bA = torch.from_numpy(synth.hash_normal((CA,), seed + 2))
    Cc = CA + C2
    wB = torch.from_numpy(synth.hash_normal((CB, Cc, ksB, ksB), seed + 3)) / np.sqrt(Cc * ksB * ksB)
    bB = torch.from_numpy(synth.hash_normal((CB,), seed + 4))
    gamma = torch.from_numpy(synth.hash_uniform((Cc,), seed + 5)) + 0.5
    beta = torch.from_numpy(synth.hash_normal((Cc,), seed + 6)) * 0.2
    skip = torch.from_numpy(synth.hash_normal((B, C2, 2 * Hs, 2 * Ws), seed + 7)) * 0.8 if C2 else None
    groups = ou.gn_groups(Cc)
    mid = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wA, bA, padding=1)
    h = torch.cat([mid, skip], 1) if C2 else mid
    if act:
        h = F.group_norm(h, groups, gamma, beta, eps=1e-5)
    if act == 2:
        h = F.silu(h)
    want = F.conv2d(h, wB, bB, padding=ksB // 2)
    d_mid = torch.full(tuple(mid.shape), float("nan"), device=DEV)
    d_out = torch.full(tuple(want.shape), float("nan"), device=DEV)
    xd = x.to(DEV)
    sd = skip.to(DEV) if C2 else None
    arrs = [np.ascontiguousarray(t.numpy(), dtype=np.float32) for t in (wA, bA, gamma, beta, wB, bB)]
    used = ctypes.c_int32(-1)
    _lib.call("ipdm_op_up_conv_chain", _lib.ptr(xd), C, B, Hs, Ws, _lib.ptr(arrs[0]), _lib.ptr(arrs[1]), CA, _lib.ptr(sd), C2,
              groups, _lib.ptr(arrs[2]), _lib.ptr(arrs[3]), act, _lib.ptr(arrs[4]), _lib.ptr(arrs[5]), CB, ksB, _lib.ptr(d_mid),
              _lib.ptr(d_out), ctypes.byref(used), _lib.current_stream())
    wide_mfma = _lib.lib().ipdm_conv_layout_code(CA, 3, 1) in (2, 4)
    # 3: the F(2x2,2x2) form of the parity convolutions (conv_wup2.hip: whole 128-cout tiles, 16-channel chunks)
    wup2 = wide_mfma and CA % 128 == 0 and C % 16 == 0 and C >= 32 and not _lib.get_option("conv_no_wup2")
    assert used.value == (3 if wup2 else 1 if wide_mfma else (2 if 4 < CA <= 16 else 0)), used.value
    assert (d_mid.cpu() - mid).abs().max() <= 2e-5 * max(1.0, mid.abs().max().item())
    err = (d_out.cpu() - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), (err, case)
    return d_mid.cpu(), d_out.cpu()


@pytest.mark.parametrize("case", UP2_CASES)
def test_upsample_conv_parity_form(case):
    _up_conv_chain(case)


def test_upsample_winograd_form_against_the_2x2_tap_form():
    """conv_wup2 (the four parity convolutions in the Winograd F(2x2,2x2) domain) and conv_ws's 2x2-tap parity kernel are the same
    function in exact arithmetic: both within the tolerance of the torch reference (inside _up_conv_chain, which also asserts which
    kernel ran) and within float32 rounding of each other, fused statistics and every edge kind included."""
    from ipdm_pytorch_amd import _lib
    n = 0
    for case in UP2_CASES:
        B, C, Hs, Ws, CA = case[:5]
        if CA % 128 or C % 16 or C < 32:
            continue
        mid, out = _up_conv_chain(case)
        with _lib.option("conv_no_wup2", 1):
            mid0, out0 = _up_conv_chain(case)
        assert not torch.equal(mid, mid0)          # (another summation order: the option really switches kernels)
        assert (mid - mid0).abs().max() <= 1.5e-5 * max(1.0, mid0.abs().max().item()), case
        assert (out - out0).abs().max() <= 1.5e-5 * max(1.0, out0.abs().max().item()), case
        n += 1
    assert n >= 8


def test_fused_statistics_equal_activation_pass():
    """The two ways of forming GroupNorm statistics (fused per-tile partial sums / option gn_unfused: a pass over the
    activations) agree to float32 rounding through a whole small UNet, concat inputs and materialised concats included."""
    from ipdm_pytorch_amd import _lib
    net, _ = _native_unet(SMALL_CFGS["b"], 11)
    x = torch.from_numpy(synth.hash_normal(SMALL_SHAPES["b"], 101)).to(DEV)
    a = net(x, 7).cpu()
    with _lib.option("gn_unfused", 1):
        b = net(x, 7).cpu()
    assert (a - b).abs().max() <= 5e-6 and float(a.abs().max()) > 0.1


def test_layout_option_changed_under_a_live_handle_is_an_error():
    """A switch that shapes packed weights / kernel choice (here conv_legacy) flipped between ipdm_unet_create and a forward
    must fail loudly, not run kernels on a layout packed for other ones; per-call switches may change."""
    from ipdm_pytorch_amd import _lib
    net, _ = _native_unet(SMALL_CFGS["a"], 11)
    x = torch.from_numpy(synth.hash_normal(SMALL_SHAPES["a"], 101)).to(DEV)
    a = net(x, 3).clone()
    with _lib.option("conv_legacy", 1):
        with pytest.raises(_lib.IpdmError, match="changed after ipdm_unet_create"):
            net(x, 3)
    with _lib.option("conv_no_up2", 1):            # per call: allowed (both weight sets are packed)
        net(x, 3)
    assert torch.equal(net(x, 3), a)


@pytest.mark.parametrize("tag", ["a", "b", "d"])
def test_unet_orientation_equivalence(tag, golden):
    """The executor may run a forward on spatially transposed activations (3x3 kernels transposed too) when that pads the
    MFMA tiling less (2000x912 sinograms); both orientations must reproduce the reference's outputs (unet_small.npz) and
    agree with each other to float32 rounding.  Odd, non-square sizes (23x19), up-sampling to explicit sizes, stride 2."""
    g = golden("unet_small")
    net, _ = _native_unet(SMALL_CFGS[tag], 11)
    x = torch.from_numpy(synth.hash_normal(SMALL_SHAPES[tag], 101)).to(DEV)
    outs = []
    from ipdm_pytorch_amd import _lib
    for flag in (0, 1):
        with _lib.option("unet_transpose", flag):
            got = net(x, 7).cpu().numpy()
        np.testing.assert_allclose(got, g["%s_t7" % tag], rtol=0, atol=1e-5)
        outs.append(got)
    assert not np.array_equal(outs[0], outs[1]) or True          # (summation order differs; equality is not required)
    assert np.abs(outs[0] - outs[1]).max() <= 5e-6


def test_lambda_ratio_kernel_body_golden(gd5, golden):
    """ipdm_lambda_ratio against the reference's OWN condition_lambda_ratio_cuda body (misc.npz: executed per simulated
    thread under a stub cuda.grid by tests/golden/make_golden.py) + the host clip [0.05, 0.99] (Model/model.py:558).
    float64 pow on both sides, float32 store: 1 ulp of float32 at most."""
    g = golden("misc")
    lam = torch.from_numpy(g["lambda_in"]).to(DEV)
    n = 0
    for key in g.files:
        if key.startswith("lambda_clip_"):
            i, ts = int(key.split("_i")[1].split("_")[0]), int(key.split("_ts")[1])
            got = gd5.lambda_ratio(lam, i, ts).cpu().numpy()
            assert np.abs(got - g[key]).max() <= 1.2e-7, key
            n += 1
    assert n == 5


# =========================================================================== UNet
def _native_unet(kw, seed):
    from ipdm_pytorch_amd.unet import UNetModel
    net = UNetModel(**kw).to(DEV)
    sd = synth.synth_state_dict(net._shapes, seed=seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return net, {k: torch.from_numpy(v) for k, v in sd.items()}


def test_unet_channel_counts_off_the_chunk_size_with_a_poisoned_workspace():
    """Levels of 20, 60 and 64 channels: wide 3x3 layers whose input channel counts (20, 60, 124 = 64 + 60 after a concat)
    are NOT multiples of the 8-channel K chunk, so their prologue reads GroupNorm scale / shift entries past
    [B, Cin] -- inside the arrays, where no layer's finalize writes.  The workspace is filled with NaN bit patterns before
    the forward: the result must still match the CPU oracle (the executor zeroes both arrays once per forward), and the
    Winograd switch must not matter (conv_no_wino: the direct kernel stays the covered fallback of those layers)."""
    from ipdm_pytorch_amd import _lib
    kw = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(), channel_mult=(0.3125, 0.9375, 1), num_heads=1)
    net, sd = _native_unet(kw, 17)
    x = torch.from_numpy(synth.hash_normal((2, 1, 40, 72), 911))
    want = ou.unet_forward(ou.UNetConfig(**kw), sd, x, 5)
    outs = []
    for no_wino in (0, 1):
        with _lib.option("conv_no_wino", no_wino):
            net.workspace(2, 40, 72).fill_(0xFF)                 # every float of the arena a NaN
            got = net(x.to(DEV), 5).cpu()
        assert bool(torch.isfinite(got).all())
        assert (got - want).abs().max() <= 1e-5 * max(1.0, float(want.abs().max())), float((got - want).abs().max())
        outs.append(got)
    assert (outs[0] - outs[1]).abs().max() <= 1e-5


@pytest.mark.parametrize("tag", ["a", "b", "d"])
def test_unet_small_golden(tag, golden):
    g = golden("unet_small")
    net, sd = _native_unet(SMALL_CFGS[tag], 11)
    assert list(net._shapes.keys()) == list(g[tag + "_keys"])
    x = torch.from_numpy(synth.hash_normal(SMALL_SHAPES[tag], 101))
    for t in (0, 7):
        got = net(x.to(DEV), torch.full((1,), t, dtype=torch.long)).cpu().numpy()
        np.testing.assert_allclose(got, g["%s_t%d" % (tag, t)], rtol=0, atol=1e-5)


def test_unet_head_dim_unsupported_fails_loudly():
    from ipdm_pytorch_amd import IpdmError
    net, _ = _native_unet(SMALL_CFGS["c"], 11)
    with pytest.raises(IpdmError):
        net(torch.zeros(SMALL_SHAPES["c"], device=DEV), 3)


@pytest.mark.parametrize("which", ["img", "proj"])
def test_unet_full_size_vs_oracle(which):
    """The two production UNets at reduced spatial size (oracle time) incl. odd sizes for proj."""
    if which == "img":
        kw = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(8, 16),
                  channel_mult=(1, 1, 2, 2, 4, 4))
        shape = (2, 1, 128, 128)
    else:
        kw = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(16, 32),
                  channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4))
        shape = (1, 1, 250, 114)       # -> 125x57 -> 63x29 -> 32x15 -> 16x8 ... odd sizes on the way
    net, sd = _native_unet(kw, 5)
    cfg = ou.UNetConfig(**kw)
    x = torch.from_numpy(synth.hash_normal(shape, 400))
    got = net(x.to(DEV), 11).cpu()
    want = ou.unet_forward(cfg, sd, x, 11)
    err = (got - want).abs().max().item()
    assert err <= 5e-5 * max(1.0, want.abs().max().item()), err


def test_fused_shortcut_of_the_narrow_levels_equals_its_own_launch():
    """On the 4/8/16-channel levels the 1x1 shortcut of a ResidualBlock whose channel count changes rides on the block's
    second 3x3 convolution as extra K chunks (conv_direct.hip, ConvArgs::sk_*); option direct_no_skip_fuse launches it on
    its own as the reference does (Model/model.py:116-130).  Same function, one rounding less: the production proj UNet
    (NCHW and parity-planar block inputs, both orientations) agrees to 1e-5 relative, and each form with the oracle."""
    from ipdm_pytorch_amd import _lib
    kw = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(16, 32),
              channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4))
    net, sd = _native_unet(kw, 5)
    cfg = ou.UNetConfig(**kw)
    for shape in ((2, 1, 256, 128), (1, 1, 250, 114)):
        x = torch.from_numpy(synth.hash_normal(shape, 410))
        want = ou.unet_forward(cfg, sd, x, 3)
        outs = []
        for tr in (0, 1):
            for off in (0, 1):
                with _lib.option("unet_transpose", tr), _lib.option("direct_no_skip_fuse", off):
                    outs.append(net(x.to(DEV), 3).cpu())
        sc = max(1.0, want.abs().max().item())
        for o in outs:
            assert (o - want).abs().max().item() <= 5e-5 * sc
        assert (outs[0] - outs[1]).abs().max().item() <= 1e-5 * sc and (outs[2] - outs[3]).abs().max().item() <= 1e-5 * sc
        assert not torch.equal(outs[0], outs[1])       # (the two forms really are different launches)


# =========================================================================== sampler loops
def test_guided_reverse_process_golden(golden):
    from ipdm_pytorch_amd.diffusion import GaussianDiffusion, InjectedNoise
    g = golden("loops")
    net, _ = _native_unet(LOOP_CFG, 41)
    for tag, (mode, shape, power, kw) in LOOP_CASES.items():
        gd = GaussianDiffusion(1000, "cosine", power)
        img = (torch.from_numpy(synth.hash_uniform(shape, 42)) * 0.05 + 0.17) if mode == "img" else \
            torch.from_numpy(synth.hash_uniform(shape, 43)) * 0.6
        ldct = torch.from_numpy(synth.hash_uniform(shape, 44)) * 0.05 + 0.17
        nd = int(g[tag + "_ndraws"])
        draws = [torch.from_numpy(synth.hash_normal(shape, 45 * 1000 + k)) for k in range(nd)]
        noise = InjectedNoise(draws)
        res, _, _ = gd.guided_reverse_process(
            model=net, img=img.to(DEV), mode=mode, lambda_curve=None, ldct=ldct.to(DEV), kernel_size_img=4,
            amplitude_img=30, kernel_size_proj=4, amplitude_proj=7, only_convertor=False, normal=False,
            noise_strength=None, noise=noise, **kw)
        assert noise.draw == nd
        got = np.stack([r.cpu().numpy() for r in res])
        assert got.shape == g[tag].shape
        np.testing.assert_allclose(got, g[tag], rtol=0, atol=5e-5)


def test_adaptive_pass_schedule_golden(golden):
    """t_start=None (Model/model.py:532-536,582-613,639-640) through the C ABI: branch taken,
    noise_strength returned, draws consumed and iterates against the reference's own run."""
    from ipdm_pytorch_amd.diffusion import GaussianDiffusion, InjectedNoise
    from tests.golden.cases import ADAPT_CASES
    g = golden("adaptive")
    net, _ = _native_unet(LOOP_CFG, 41)
    for tag, (mode, shape, power, amp, ns_in, kw) in ADAPT_CASES.items():
        gd = GaussianDiffusion(1000, "cosine", power)
        img = (torch.from_numpy(synth.hash_uniform(shape, 42)) * 0.05 + 0.17) if mode == "img" else \
            torch.from_numpy(synth.hash_uniform(shape, 43)) * 0.6
        ldct = torch.from_numpy(synth.hash_uniform(shape, 44)) * 0.05 + 0.17
        nd = int(g[tag + "_ndraws"])
        noise = InjectedNoise([torch.from_numpy(synth.hash_normal(shape, 48 * 1000 + k)) for k in range(nd)])
        res, _, ns = gd.guided_reverse_process(
            model=net, img=img.to(DEV), t_start=None, mode=mode, lambda_curve=None, ldct=ldct.to(DEV), kernel_size_img=4,
            amplitude_img=amp, kernel_size_proj=4, amplitude_proj=amp, only_convertor=False, normal=False,
            noise_strength=ns_in, constant_guidance=None, noise=noise, **kw)
        assert str(ns) == str(g[tag + "_ns"]), tag
        assert noise.draw == nd, tag
        got = np.stack([r.cpu().numpy() for r in res])
        assert got.shape == g[tag].shape
        np.testing.assert_allclose(got, g[tag], rtol=0, atol=1e-4, err_msg=tag)


def test_sparse_guided_reverse_process_golden(golden):
    """sample_method='sparse' (DDIM, Model/model.py:654-759) through the C ABI against the reference's outputs."""
    from ipdm_pytorch_amd.diffusion import GaussianDiffusion, InjectedNoise, NoiseSource
    from tests.golden.cases import SPARSE_CASES
    g = golden("sparse")
    net, _ = _native_unet(LOOP_CFG, 41)
    for tag, (shape, power, kw) in SPARSE_CASES.items():
        gd = GaussianDiffusion(1000, "cosine", power)
        cond = torch.from_numpy(synth.hash_uniform(shape, 46)) * 0.6
        nd = int(g[tag + "_ndraws"])
        noise = InjectedNoise([torch.from_numpy(synth.hash_normal(shape, 47 * 1000 + k)) for k in range(nd)])
        res = gd.sparse_guided_reverse_process(model=net, condition=cond.to(DEV), noise=noise, **kw)
        assert noise.draw == nd                                  # one draw per q_sample and per DDIM step, as torch.randn_like
        got = np.stack([r.cpu().numpy() for r in res])
        assert got.shape == g[tag].shape
        np.testing.assert_allclose(got, g[tag], rtol=0, atol=5e-5)
    # batch of two slices == two single-slice runs (per-slice statistics)
    shape, power, kw = SPARSE_CASES["img"]
    gd = GaussianDiffusion(1000, "cosine", power)
    cond2 = torch.from_numpy(synth.hash_uniform((2,) + shape[1:], 48)) * 0.6
    full = gd.sparse_guided_reverse_process(model=net, condition=cond2.to(DEV), noise=NoiseSource(5, 0), **kw)
    for b in range(2):
        one = gd.sparse_guided_reverse_process(model=net, condition=cond2[b:b + 1].to(DEV), noise=NoiseSource(5, b), **kw)
        for k in range(len(full)):
            assert torch.equal(full[k][b:b + 1], one[k])


def test_guided_reverse_process_batch_equals_per_slice():
    """Per-slice semantics + shard invariance: B=3 in one call == three B=1 calls (bit-identical)."""
    from ipdm_pytorch_amd.diffusion import GaussianDiffusion, NoiseSource
    net, _ = _native_unet(LOOP_CFG, 41)
    gd = GaussianDiffusion(1000, "cosine", 5)
    shape = (3, 1, 40, 24)
    img = (torch.from_numpy(synth.hash_uniform(shape, 501)) * torch.tensor([0.6, 2.0, 0.1]).view(3, 1, 1, 1)).to(DEV)
    kw = dict(model=net, t_start=[3, 2, 2], clip=False, lambda_ratio=1, eta=0.5, mode="proj", constant_guidance=None,
              kernel_size_proj=4, amplitude_proj=7, only_convertor=False, normal=False)
    full, _, _ = gd.guided_reverse_process(img=img, noise=NoiseSource(3, 0), **kw)
    for b in range(3):
        one, _, _ = gd.guided_reverse_process(img=img[b:b + 1], noise=NoiseSource(3, b), **kw)
        for k in range(len(full)):
            assert torch.equal(full[k][b:b + 1], one[k]), (b, k)


# =========================================================================== end to end
# The heavy end-to-end tests are PAIRS (tests/_oracle_pool.py, tests/conftest.py): `*_device_run` (oracle_submit, collected
# first) runs the device side and hands one CPU replay per slice / seed / precision to the session's oracle pool; the verdict
# test of the same name as in rounds 1-5 (oracle_join, collected last) joins them and asserts.  A verdict test run on its own
# (-k) performs its device run first.
def _once(pool, tag, fn):
    if tag not in pool.stash:
        pool.stash[tag] = fn(pool)
    return pool.stash[tag]


# What the default suite replays is sized by the GPU boxes' CPU quota -- SIXTEEN CPUs of time (cpu.max), not the 256 logical CPUs
# they show (tests/_oracle_pool.py): the headline-length replay alone is ~300 s of those sixteen.  Default: the headline slice,
# three full-size seeds in float32 (one of them in float64 too), three reduced seeds in float32 + float64 with every stage kept.
# IPDM_PARITY_FULL=1: the round-5 statistics -- five full-size seeds and seven reduced ones, each in float32 AND float64 -- same
# tests, same code, ~25 min on such a box (profiles/r06_parity_full.log holds this round's run).
PARITY_FULL = os.environ.get("IPDM_PARITY_FULL") == "1"

# (noise seed, weight seed of the proj net [img: + 1], phantom); the first one is ipdm_pytorch_amd.denoiser.smoke_pipeline's
REDUCED_SEEDS = ((11, 21, 1), (29, 102, 2), (43, 104, 3), (61, 106, 4), (83, 108, 5), (97, 110, 6), (113, 112, 7))[:7 if PARITY_FULL else 3]
N_STAGE_SEEDS = 3        # ... of which the first three keep every stored iterate (stage-by-stage arbiter)


def _reduced_submit(pool):
    """The reduced pipeline (true geometry, 16-channel networks; 2+2 proj steps, FBP, sharpen, 2 img steps, ultra pass) for every
    seed of REDUCED_SEEDS on the device, each replayed by the CPU oracle in float32 and float64; ONE set of replays serves the three
    reduced-pipeline tests (PSNR of the canonical seed, stage-by-stage arbiter, arbiter over the seeds)."""
    from tests import _oracle_child as oc
    runs = []
    for k, (seed, wp, ph) in enumerate(REDUCED_SEEDS):
        stages = k < N_STAGE_SEEDS
        den, opt, sino = _reduced_denoiser(seed, wp, wp + 1, ph, save_it_state_proj=stages, save_it_state_img=stages)
        out = den.progressive_denoiser(save_proj_state=stages, sharpen_num=70).cpu().numpy()
        draws = [z.cpu().numpy() for z in den.noise.draws]
        hs = {}
        for dt, thr in (("float64", 3), ("float32", 2)):
            job = pool.path("reduced%d_%s.npz" % (seed, dt))
            oc.write_job(job, opt.__dict__, sino, draws, wp, 70, dtype=dt, nets="smoke", mid=stages)
            hs[dt] = pool.submit("reduced seed %d %s" % (seed, dt), job, thr)
        hip = None
        if stages:
            n_p = len(den.proj_denoise_result)
            hip = ([np.array(den.proj_denoise_result[j + 1]) for j in range(n_p)], np.array(den.proj_denoise_convert2img_result[n_p]),
                   [np.array(den.progressive_denoise_result[j + 1]) for j in range(len(den.progressive_denoise_result))])
        runs.append(dict(seed=seed, out=out, hip=hip, h=hs))
    return runs


@pytest.mark.oracle_join
def test_smoke_pipeline_matches_oracle_psnr(oracle_pool):
    """proj GRP -> FBP -> sharpen -> img GRP -> ultra on a real-geometry phantom sinogram, reduced UNets.
    north_star: PSNR (vs ground truth, on miu2pixel images) within 1e-4 relative of the CPU path."""
    from ipdm_pytorch_amd.denoiser import smoke_pipeline
    run = _once(oracle_pool, "reduced", _reduced_submit)[0]          # REDUCED_SEEDS[0] IS smoke_pipeline's configuration ...
    got, want = run["out"], oracle_pool.result(run["h"]["float32"])
    assert np.array_equal(smoke_pipeline(DEV)[0], got)              # ... bit for bit (what __graft_entry__.smoke() runs)
    assert got.shape == want.shape == (1, 1, 512, 512)
    # Eleven evaluations of random-weight networks amplify float32 rounding ~100x: the float32 CPU oracle itself ends
    # 2.4e-4 (max-abs; rms 3.7e-6) from the float64 value of the same function (test_smoke_pipeline_fp64_arbiter, which
    # bounds this library's distance to that value by 1.5x the oracle's).  Two float32 evaluations may therefore differ by
    # the sum, E2E_MAX_REL; the rms sits two orders of magnitude below; north_star's acceptance metric is the PSNR below.
    err = np.abs(got - want)
    assert err.max() <= E2E_MAX_REL * max(1.0, np.abs(want).max()), float(err.max())
    assert np.sqrt((err.astype(np.float64) ** 2).mean()) <= 2e-5, float(np.sqrt((err.astype(np.float64) ** 2).mean()))
    truth = od.miu2pixel(torch.from_numpy(synth.rasterize(synth.ellipse_phantom(1)))).numpy()
    p_hip = od.psnr(truth, od.miu2pixel(torch.from_numpy(got[0, 0])).numpy())
    p_cpu = od.psnr(truth, od.miu2pixel(torch.from_numpy(want[0, 0])).numpy())
    assert abs(p_hip - p_cpu) <= 1e-4 * abs(p_cpu), (p_hip, p_cpu)


def _reduced_denoiser(seed, wp, wi, phantom, **over):
    """The reduced (smoke) networks with weight seeds (wp, wi) behind the drop-in surface, one real-geometry phantom loaded,
    the device's draws recorded."""
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, SMOKE_PROJ, SMOKE_IMG, _RecordingNoise
    from ipdm_pytorch_amd.diffusion import NoiseSource
    from ipdm_pytorch_amd.unet import UNetModel
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(dict(device=DEV, t_start_proj=[2, 2], t_start_img=[2], ultra_img_denoise=True), **over), opt.__dict__)
    den = progressive_domain_denoiser(opt, seed=seed)
    den.proj_model = UNetModel(**SMOKE_PROJ).to(DEV)
    den.img_model = UNetModel(**SMOKE_IMG).to(DEV)
    den.proj_model.load_state_dict({n: torch.from_numpy(v) for n, v in synth.synth_state_dict(den.proj_model._shapes, seed=wp).items()})
    den.img_model.load_state_dict({n: torch.from_numpy(v) for n, v in synth.synth_state_dict(den.img_model._shapes, seed=wi).items()})
    sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(phantom)), seed=phantom)
    den.data_sample_load(ldproj=torch.from_numpy(sino)[None, None])
    den.noise = _RecordingNoise(NoiseSource(seed, 0))
    return den, opt, sino


@pytest.mark.oracle_join
def test_smoke_pipeline_fp64_arbiter(oracle_pool):
    """Who is right when two float32 evaluations differ?  The reduced end-to-end pipeline once more on the CPU in FLOAT64
    (same float32 inputs, weights, draws and schedule constants: oracle.pipeline.progressive_slice on float64 tensors) is
    the value both approximate.  STAGE BY STAGE -- every stored iterate of the projection loop, the FBP image, every iterate of
    the image loops -- for THREE seeds (weights, phantom, dose noise, draws all vary).  Up to the first amplifying pass both
    evaluations sit at 1e-7 of the arbiter and the comparison is sharp for every seed (HIP at most 1.5x as far as the float32
    CPU oracle, rms and max-abs).  One image-domain pass of these random-weight networks then amplifies rounding ~100x,
    chaotically (round 3: the oracle's own distance moved 2.4x with nothing but its thread count): after it a single seed is
    one draw from that distribution, so each amplified stage is judged by the MEDIAN over the seeds (ARBITER_MEDIAN_*, the
    criterion of test_smoke_pipeline_fp64_arbiter_over_seeds) plus the hard cap on the worst one."""
    def dist(a, b):
        e = np.abs(np.asarray(a, dtype=np.float64).reshape(np.asarray(b).shape) - np.asarray(b, dtype=np.float64))
        return float(e.max()), float(np.sqrt((e ** 2).mean()))
    per_stage = {}                       # stage -> [(ratio max-abs, ratio rms, amplified)] over the seeds
    for k, run in enumerate(_once(oracle_pool, "reduced", _reduced_submit)[:N_STAGE_SEEDS]):
        seed, hip = run["seed"], run["hip"]
        (_, z64), (_, z32) = oracle_pool.result(run["h"]["float64"], mid=True), oracle_pool.result(run["h"]["float32"], mid=True)
        m64, m32 = (z64["proj"], z64["fbp"], z64["img"]), (z32["proj"], z32["fbp"], z32["img"])
        n_p, n_i = len(m64[0]), len(m64[2])
        assert len(hip[0]) == n_p and len(hip[2]) == n_i
        stages = [("proj iter_%d" % (j + 1), lambda m, j=j: m[0][j]) for j in range(n_p)] + [("fbp", lambda m: m[1])] + \
                 [("img iter_%d" % (j + 1), lambda m, j=j: m[2][j]) for j in range(n_i)]
        for name, pick in stages:
            h, c = dist(pick(hip), pick(m64)), dist(pick(m32), pick(m64))
            amplified = c[1] > 1e-6
            print("fp64 arbiter seed %d %-11s |hip-f64| max %.3e rms %.3e | |cpu32-f64| max %.3e rms %.3e | ratio max %.2f rms %.2f%s"
                  % (seed, name, h[0], h[1], c[0], c[1], h[0] / c[0], h[1] / c[1], "  (amplified)" if amplified else ""))
            per_stage.setdefault(name, []).append((h[0] / c[0], h[1] / c[1], amplified))
            if not amplified:            # sharp, seed by seed
                assert h[1] <= 1.5 * c[1] and h[0] <= 1.5 * c[0], (seed, name, h, c)
            else:
                assert h[1] <= ARBITER_WORST_RMS * c[1] and h[0] <= ARBITER_WORST_MAX * c[0], (seed, name, h, c)
        # float32 vs float32 at the end of the chain: bounded by what the arbiter justifies (the sum of the two distances).  The
        # frozen bound is the canonical seed's; other weight seeds of the reduced networks are judged by the arbiter only.
        if k == 0:
            ff = dist(hip[2][-1], m32[2][-1])
            scale = float(np.abs(m64[2][-1]).max())
            print("fp64 arbiter: |hip-cpu32| at the end max %.3e rms %.3e (scale %.3f)" % (ff[0], ff[1], scale))
            assert ff[0] <= E2E_MAX_REL * max(1.0, scale)
    for name, rs in per_stage.items():
        amp = [r for r in rs if r[2]]
        if len(amp) >= 2:                # the stages behind the amplifying pass: the median over the seeds is the measurement
            med_max, med_rms = float(np.median([r[0] for r in amp])), float(np.median([r[1] for r in amp]))
            print("fp64 arbiter %-11s median over %d seeds: max-abs ratio %.2f rms ratio %.2f" % (name, len(amp), med_max, med_rms))
            assert med_rms <= ARBITER_MEDIAN_RMS and med_max <= ARBITER_MEDIAN_MAX, (name, rs)


@pytest.mark.oracle_join
def test_smoke_pipeline_fp64_arbiter_over_seeds(oracle_pool):
    """The fp64 arbiter as a statistic: the reduced end-to-end pipeline (proj loop 2+2 steps, FBP, sharpen, img loop, ultra
    pass) over the seeds of REDUCED_SEEDS (three by default, SEVEN under IPDM_PARITY_FULL=1) -- network weights, phantom, dose
    noise and diffusion draws all vary -- each replayed by the CPU oracle in float32 and in float64 (pinned child processes).
    Median over the seeds of err(HIP, fp64) / err(oracle32, fp64) at the END of the chain (after the amplifying image-domain
    passes): <= 1.25 in rms, <= 1.5 in max-abs (asserted with five seeds or more; the hard caps on the worst seed always)."""
    runs = _once(oracle_pool, "reduced", _reduced_submit)
    hips = [r["out"] for r in runs]
    c32s, f64s = [oracle_pool.result(r["h"]["float32"]) for r in runs], [oracle_pool.result(r["h"]["float64"]) for r in runs]
    for h, c in zip(hips, c32s):
        print("reduced pipeline: |hip - cpu32| max %.3e rms %.3e (scale %.3f)" % (np.abs(h - c).max(), np.sqrt(((h - c).astype(np.float64) ** 2).mean()), np.abs(c).max()))
    _arbiter_ratios("reduced pipeline", hips, c32s, f64s)


def test_pipeline_is_bit_reproducible():
    """No float atomics anywhere on the path (fixed-order GroupNorm and step statistics, in-order back-projection,
    counter-based noise): two runs of the reduced end-to-end pipeline give the same bits."""
    from ipdm_pytorch_amd.denoiser import smoke_pipeline
    a, _ = smoke_pipeline(DEV)
    b, _ = smoke_pipeline(DEV)
    assert np.array_equal(a, b)


def test_drop_in_surface():
    """update_opt / reset_opt / result dicts behave as the reference's (Utils/train_test_utils.py:202-211,45-56)."""
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, ResultTempDict
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    den = progressive_domain_denoiser(opt)
    den.update_opt(dict(convertor="FBP", save_it_state_img=False, ultra_img_denoise=False, not_a_key=1))
    assert den.opt.ultra_img_denoise is False and not hasattr(den.opt, "not_a_key")
    den.reset_opt()
    assert den.opt.ultra_img_denoise is True
    with pytest.raises(AttributeError):
        den.update_opt(None)
    d = ResultTempDict()
    d["iter_1"], d["iter_2"] = 1, 2
    assert d[1] == 1 and d[-1] == 2 and d["iter_2"] == 2
    den.update_opt(dict(benchmark_test=True))
    x = torch.zeros((1, 1, 2000, 912))
    den.data_sample_load(ldproj=x)
    out, ns = den.proj_denoiser(den.ldproj, save_state=False)      # only_convertor short-circuit: FBP of the input
    assert isinstance(out, torch.Tensor) and out.device.type == "cpu" and tuple(out.shape) == (1, 1, 512, 512) and ns is None


def test_drop_in_sparse_sample_method():
    """update_opt(sample_method_*='sparse') routes both domains through the DDIM sampler with the reference's
    hard-wired guidance ranges (Utils/train_test_utils.py:445-453,505-514) and keeps the result-dict conventions."""
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, SMOKE_PROJ, SMOKE_IMG
    from ipdm_pytorch_amd.unet import UNetModel
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    den = progressive_domain_denoiser(opt, seed=3)
    den.proj_model = UNetModel(**SMOKE_PROJ).to(DEV)
    den.img_model = UNetModel(**SMOKE_IMG).to(DEV)
    for m, seed in ((den.proj_model, 21), (den.img_model, 22)):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(m._shapes, seed=seed).items()})
    den.update_opt(dict(sample_method_proj="sparse", sample_method_img="sparse", t_start_proj=[4, 3], ddim_timesteps_proj=[2, 1],
                        t_start_img=[3, 3], ddim_timesteps_img=[1, 2], ultra_img_denoise=False))
    sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(2)), seed=2)
    den.data_sample_load(ldproj=torch.from_numpy(sino)[None, None])
    out = den.progressive_denoiser(sharpen_num=70)
    assert tuple(out.shape) == (1, 1, 512, 512) and bool(torch.isfinite(out).all())
    assert len(den.progressive_denoise_result) == 1 and den.noise_strength is None
    den.update_opt(dict(sample_method_img="nonsense"))
    with pytest.raises(ValueError):
        den.img_denoiser(out)


def test_drop_in_adaptive_schedule_hand_off():
    """t_start_proj=None / t_start_img=None (Utils/train_test_utils.py:553-566): the proj pass picks the branch from the
    Delta map, reports noise_strength, and the img pass takes its pass list from that hand-off (4 iterates each:
    3 adaptive passes + the final average, the t=20 probe pass dropped, Model/model.py:639-640)."""
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, SMOKE_PROJ, SMOKE_IMG
    from ipdm_pytorch_amd.unet import UNetModel
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    den = progressive_domain_denoiser(opt, seed=5)
    den.proj_model = UNetModel(**SMOKE_PROJ).to(DEV)
    den.img_model = UNetModel(**SMOKE_IMG).to(DEV)
    for m, seed in ((den.proj_model, 21), (den.img_model, 22)):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(m._shapes, seed=seed).items()})
    calls = {"proj": 0, "img": 0}

    class counted:
        def __init__(self, net, name):
            self.net, self.name = net, name

        def __call__(self, x, t):
            calls[self.name] += 1
            return self.net(x, t)

        def __getattr__(self, k):
            return getattr(self.net, k)
    den.proj_model, den.img_model = counted(den.proj_model, "proj"), counted(den.img_model, "img")
    den.update_opt(dict(t_start_proj=None, t_start_img=None, constant_guidance_img=None, ultra_img_denoise=False,
                        save_it_state_proj=True, save_it_state_img=True))
    sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(2)), seed=2)
    den.data_sample_load(ldproj=torch.from_numpy(sino)[None, None])
    out = den.progressive_denoiser(sharpen_num=70, save_proj_state=True)
    assert tuple(out.shape) == (1, 1, 512, 512) and bool(torch.isfinite(out).all())
    want_proj = {"high": 20 + 75, "mid": 20 + 53, "low": 20 + 45}
    want_img = {"high": 20 + 45, "mid": 20 + 37, "low": 20 + 30}
    assert den.noise_strength in want_proj
    assert calls["proj"] == want_proj[den.noise_strength] and calls["img"] == want_img[den.noise_strength], calls
    assert len(den.proj_denoise_result) == 4 and len(den.proj_denoise_convert2img_result) == 4
    assert len(den.progressive_denoise_result) == 4


def test_normal_mode_reports_inverse_transformed_iterates():
    """opt.normal (Model/model.py:616-617): the loop runs on the transformed tensor and every reported iterate (and
    their final average) goes through the inverse power transform; nothing else changes."""
    from ipdm_pytorch_amd.diffusion import GaussianDiffusion, NoiseSource
    from ipdm_pytorch_amd.normalize import yeo_johnson_transform, yeo_johnson_inverse_transform
    net, _ = _native_unet(LOOP_CFG, 41)
    gd = GaussianDiffusion(1000, "cosine", 5)
    raw = torch.from_numpy(synth.hash_uniform((2, 1, 40, 24), 43)) * 0.6
    x, trs = yeo_johnson_transform(raw)
    x = x.to(torch.float32).to(DEV)
    kw = dict(model=net, img=x, mode="proj", t_start=[3, 2], clip=False, lambda_ratio=1, eta=0.5, constant_guidance=None,
              lambda_curve=None, kernel_size_proj=4, amplitude_proj=7, only_convertor=False, noise_strength=None)
    plain, _, _ = gd.guided_reverse_process(normal=False, noise=NoiseSource(7, 0), **kw)
    norm, _, _ = gd.guided_reverse_process(normal=True, transformer=trs, noise=NoiseSource(7, 0), **kw)
    assert len(norm) == len(plain) == 3
    for k in range(2):
        want = yeo_johnson_inverse_transform(plain[k], trs).to(torch.float32)
        assert torch.equal(norm[k], want)
    assert torch.allclose(norm[2], (norm[0] + norm[1]) / 2, atol=1e-6)          # the average of the REPORTED iterates


def test_drop_in_normal_option():
    """update_opt(normal=True) end to end on the reduced nets: data_sample_load transforms, both loops report through the
    inverse transforms (Utils/train_test_utils.py:560-562,578-588)."""
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, SMOKE_PROJ, SMOKE_IMG
    from ipdm_pytorch_amd.unet import UNetModel
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    den = progressive_domain_denoiser(opt, seed=6)
    den.proj_model = UNetModel(**SMOKE_PROJ).to(DEV)
    den.img_model = UNetModel(**SMOKE_IMG).to(DEV)
    for m, seed in ((den.proj_model, 21), (den.img_model, 22)):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(m._shapes, seed=seed).items()})
    den.update_opt(dict(normal=True, t_start_proj=[2, 2], t_start_img=[2], ultra_img_denoise=False))
    sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(2)), seed=2)
    den.data_sample_load(ldproj=torch.from_numpy(sino)[None, None])
    assert den.trans_ldproj is not None and abs(float(den.ldproj.mean())) < 1e-3       # standardised input
    out = den.progressive_denoiser(sharpen_num=70)
    assert tuple(out.shape) == (1, 1, 512, 512) and bool(torch.isfinite(out).all()) and den.trans_ldimg is not None


def test_drop_in_dataset_evaluation(tmp_path):
    """fit() in mode test_prog (Utils/train_test_utils.py:274-322,337-348): dataset trees on disk -> per-slice
    progressive sample -> metric.json per slice and for the run, result archives."""
    import json
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, SMOKE_PROJ, SMOKE_IMG
    from ipdm_pytorch_amd.unet import UNetModel
    root = str(tmp_path / "data")
    for k, sid in enumerate((2, 5)):
        ell = synth.ellipse_phantom(sid)
        mu = synth.rasterize(ell).astype(np.float32)
        sino = synth.fan_sinogram(ell)
        for kind, arr in (("fdimg", mu), ("ldimg", mu + 0.004 * synth.hash_normal(mu.shape, 900 + sid).astype(np.float32)),
                          ("ldproj", synth.low_dose(sino, seed=sid))):
            os.makedirs(os.path.join(root, kind, "L%03d" % sid), exist_ok=True)
            np.savez(os.path.join(root, kind, "L%03d" % sid, "slice_%03d.npz" % k), arr.astype(np.float32))
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(test_dataset_path_FD_img=root + "/fdimg", test_dataset_path_LD_img=root + "/ldimg",
                  test_dataset_path_LD_proj=root + "/ldproj", test_numbers=0, metrics=["psnr", "ssim", "nqm"],
                  test_result_data_save=True, t_start_proj=[2, 2], t_start_img=[2], ultra_img_denoise=False), opt.__dict__)
    den = progressive_domain_denoiser(opt, result_save_path=str(tmp_path / "out"), seed=4)
    den.proj_model = UNetModel(**SMOKE_PROJ).to(DEV)
    den.img_model = UNetModel(**SMOKE_IMG).to(DEV)
    for m, seed in ((den.proj_model, 21), (den.img_model, 22)):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(m._shapes, seed=seed).items()})
    den.fit()
    run = os.path.join(str(tmp_path / "out"), "IPDM_default", "save_test_results", "Save_Iter_0")
    tot = json.load(open(os.path.join(run, "metric.json")))
    assert set(tot) >= {"LDCT", "deProg"} and np.isfinite(tot["LDCT"]["psnr_iter_0"]) and "psnr_iter_0_std" in tot["LDCT"]
    assert tot["LDCT"]["psnr_iter_0"] > 30 and 0 < tot["LDCT"]["ssim_iter_0"] <= 1
    one = os.path.join(run, "L002", "slice_000")
    assert os.path.isfile(os.path.join(one, "metric.json")) and os.path.isfile(os.path.join(one, "prog_denoise_result.npz"))
    assert len(den.metric_each_sample) == 2 and "psnr_iter_1" in den.metric_each_sample[0]["deProg"]


# =========================================================================== BASELINE.json's full sizes
FULL_IMG = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(8, 16), channel_mult=(1, 1, 2, 2, 4, 4))
FULL_PROJ = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(16, 32),
                 channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4))


@pytest.mark.parametrize("which", ["img", "proj"])
def test_unet_true_size_vs_oracle(which):
    """The production UNets at their TRUE input sizes (512x512 / 2000x912; attention over T=4096 / 7125 keys,
    persistent multi-round conv schedules, the narrow-layer kernels at 2000x912) against the CPU oracle."""
    kw, shape = (FULL_IMG, (1, 1, 512, 512)) if which == "img" else (FULL_PROJ, (1, 1, 2000, 912))
    net, sd = _native_unet(kw, 6)
    x = torch.from_numpy(synth.hash_normal(shape, 401))
    got = net(x.to(DEV), 13).cpu()
    torch.set_num_threads(host_threads())      # (a quarter of the box's CPU quota: the oracle pool's replays hold the rest)
    want = ou.unet_forward(ou.UNetConfig(**kw), sd, x, 13)
    err = (got - want).abs().max().item()
    assert err <= 5e-5 * max(1.0, want.abs().max().item()), err


def _full_size_run(pool, opt_over, seed, phantoms, tag, replay=None, arbiter=False, threads=3, also_bf16x3=False):
    """The production networks at full size on the device (batch = len(phantoms), global slice ids 0..), the draws recorded,
    then the slices in `replay` (default: all) handed to the oracle pool: one float32 replay each in its own pinned child
    process (tests/_oracle_child.py), with arbiter=True a float64 one too.  Returns (device output, [f32 handles], [f64 handles])."""
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, _RecordingNoise
    from ipdm_pytorch_amd.diffusion import NoiseSource
    from tests import _oracle_child as oc
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(opt_over, device=DEV), opt.__dict__)
    den = progressive_domain_denoiser(opt, seed=seed)        # full-size UNets, deterministic synthetic weights (seed 0)
    sinos = np.stack([synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(p)), seed=p) for p in phantoms])
    den.data_sample_load(ldproj=torch.from_numpy(sinos)[:, None])
    rec = _RecordingNoise(NoiseSource(seed, 0))
    den.noise = rec
    got = den.progressive_denoiser(sharpen_num=70).cpu().numpy()
    if also_bf16x3:      # the same sample once more with the wide 3x3 layers on conv_wino3 (opt-in; the same draws: the counter-based source restarted)
        from ipdm_pytorch_amd import _lib
        den.temp_clear()
        den.noise = NoiseSource(seed, 0)
        with _lib.option("conv_bf16x3", 1):
            got = (got, den.progressive_denoiser(sharpen_num=70).cpu().numpy())
    h32, h64 = [], []
    for b in (range(len(phantoms)) if replay is None else replay):
        draws = [z[b:b + 1].cpu().numpy() for z in rec.draws]
        if arbiter:      # the same slice once more in float64: the value both float32 evaluations approximate (slowest: first)
            job = pool.path("%s_job%d_f64.npz" % (tag, b))
            oc.write_job(job, opt.__dict__, sinos[b], draws, 0, 70, dtype="float64")
            h64.append(pool.submit("%s slice %d f64" % (tag, b), job, threads + 1))
        job = pool.path("%s_job%d.npz" % (tag, b))
        oc.write_job(job, opt.__dict__, sinos[b], draws, 0, 70)
        h32.append(pool.submit("%s slice %d f32" % (tag, b), job, threads))
    del den, rec
    torch.cuda.empty_cache()
    return got, h32, h64


def _check_full_size(got, want, phantom, max_rel):
    """north_star's acceptance metric -- PSNR against the ground-truth phantom within 1e-4 relative of the CPU path on
    identical inputs and noise -- plus max-abs / rms distances (returned for the report)."""
    assert got.shape == want.shape == (1, 1, 512, 512)
    err = np.abs(got.astype(np.float64) - want)
    scale = max(1.0, float(np.abs(want).max()))
    truth = od.miu2pixel(torch.from_numpy(synth.rasterize(synth.ellipse_phantom(phantom)))).numpy()
    p_hip = od.psnr(truth, od.miu2pixel(torch.from_numpy(got[0, 0])).numpy())
    p_cpu = od.psnr(truth, od.miu2pixel(torch.from_numpy(want[0, 0])).numpy())
    assert abs(p_hip - p_cpu) <= 1e-4 * abs(p_cpu), (p_hip, p_cpu)
    rms = float(np.sqrt((err ** 2).mean()))
    # (if this ever fires with a few 1e-4 along a streak while the PSNRs above agree: one 4x4 block of the guidance map has
    #  crossed the JUMP of the reference's weight_lambda at 1.7 -- Utils/train_test_utils.py:831-865, DESIGN 4 -- in one of
    #  the two float32 evaluations; both are valid, another seed settles it)
    assert err.max() <= max_rel * scale, (float(err.max()), rms, int((err > max_rel * scale).sum()), "pixels above the bound")
    return float(err.max()), rms, p_hip, p_cpu


ARBITER_MEDIAN_RMS, ARBITER_MEDIAN_MAX = 1.25, 1.5       # median over the seeds of err(HIP, fp64) / err(oracle32, fp64)
# Hard caps on the worst seed.  Measured (round 4, gpurun_out/arbiter_*.txt): full size -- the production kernels -- rms
# ratios 0.99 ... 1.03, max-abs 0.93 ... 1.24 (nothing amplifies there: both evaluations end 1.3e-7 rms from the arbiter).  The
# reduced random-weight networks amplify rounding chaotically: over five weight seeds the float32 ORACLE's own distance to the
# arbiter spans 1.9e-7 ... 4.3e-4 rms, and the ratio of two such samples 0.86 ... 2.16 (rms), 0.75 ... 2.48 (max-abs) around
# a median of 1.00 / 1.32 -- a single seed's ratio is noise, the median is the measurement.  Round 5: seven seeds in the reduced
# statistic (median 0.89 / 1.13, profiles/r05_arbiter_reduced_pipeline.txt), the stage-by-stage test judged by its median behind
# the amplifying pass, caps 4 -> 3 in rms and 3.5 in max-abs: the worst seed (the fourth) read 2.16 / 2.48 in round 4 and
# 2.17 / 2.67 in round 5 with nothing changed on the device side -- the CPU oracle's own float32 replay moves with its host's
# thread schedule, and the max over pixels of two chaotic evaluations is the noisier of the two statistics.
ARBITER_WORST_RMS, ARBITER_WORST_MAX = 3.0, 3.5


def _arbiter_ratios(tag, hips, c32s, f64s):
    """err(HIP, fp64) / err(CPU float32 oracle, fp64) per seed, rms and max-abs, asserted as a STATISTIC over the seeds:
    a float32 evaluation of these chains ends a chaotic distance from the exact value (one seed's ratio says little), but
    over seeds a library whose kernels round worse than the reference's would sit above 1 systematically."""
    r_rms, r_max = [], []
    for h, c, f in zip(hips, c32s, f64s):
        eh, ec = np.abs(h.astype(np.float64) - f), np.abs(c.astype(np.float64) - f)
        r_rms.append(float(np.sqrt((eh ** 2).mean()) / max(np.sqrt((ec ** 2).mean()), 1e-30)))
        r_max.append(float(eh.max() / max(ec.max(), 1e-30)))
        print("fp64 arbiter %s: |hip-f64| max %.3e rms %.3e | |cpu32-f64| max %.3e rms %.3e | ratio max %.2f rms %.2f"
              % (tag, eh.max(), np.sqrt((eh ** 2).mean()), ec.max(), np.sqrt((ec ** 2).mean()), r_max[-1], r_rms[-1]))
    msg = "fp64 arbiter %s over %d seeds: rms ratio median %.2f worst %.2f | max-abs ratio median %.2f worst %.2f" % (
        tag, len(r_rms), float(np.median(r_rms)), max(r_rms), float(np.median(r_max)), max(r_max))
    print(msg)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "arbiter_%s.txt" % tag.replace(" ", "_")), "w") as f:
        f.write(msg + "\n" + "rms ratios %s\nmax-abs ratios %s\n" % (["%.3f" % x for x in r_rms], ["%.3f" % x for x in r_max]))
    if len(r_rms) >= 5:          # (the median of fewer seeds is itself a noisy draw -- single ratios of the reduced pipeline span 0.3 ... 2.7 --:
                                 #  the default suite's three seeds are held by the hard caps below and by the stage-by-stage medians of
                                 #  test_smoke_pipeline_fp64_arbiter; the statistic proper is the IPDM_PARITY_FULL=1 run)
        assert np.median(r_rms) <= ARBITER_MEDIAN_RMS and np.median(r_max) <= ARBITER_MEDIAN_MAX, msg
    assert max(r_rms) <= ARBITER_WORST_RMS and max(r_max) <= ARBITER_WORST_MAX, msg


HEADLINE_OVER = dict(t_start_proj=[15, 15, 15], t_start_img=[15], ultra_img_denoise=True)


def _headline_submit(pool):
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser
    # (the longest replay: ~64 % of the pool's threads -- nine of the fourteen a 16-CPU box leaves for replays --, at most 32)
    got, h32, _ = _full_size_run(pool, HEADLINE_OVER, 1234, [0, 1], "headline", replay=[1], threads=max(2, min(32, int(pool.capacity * 0.64 + 0.5))))
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(HEADLINE_OVER, device=DEV), opt.__dict__)
    one = progressive_domain_denoiser(opt, seed=1234)
    sino0 = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(0)), seed=0)
    one.data_sample_load(ldproj=torch.from_numpy(sino0)[None, None])
    alone = one.progressive_denoiser(sharpen_num=70).cpu().numpy()
    del one
    torch.cuda.empty_cache()
    return dict(got=got, h=h32[0], alone=alone)


FULL_SIZE_SEEDS = (17, 23, 31, 47, 59)[:5 if PARITY_FULL else 3]
FULL_SIZE_F64 = len(FULL_SIZE_SEEDS) if PARITY_FULL else 1      # how many of them are replayed in float64 too (the arbiter)


def _full_size_submit(pool):
    runs = []
    for k, seed in enumerate(FULL_SIZE_SEEDS):
        (got, got3), h32, h64 = _full_size_run(pool, dict(t_start_proj=[2, 2], t_start_img=[2], ultra_img_denoise=False), seed, [4 + k],
                                               "s%d" % seed, arbiter=k < FULL_SIZE_F64, also_bf16x3=True)
        runs.append((got, h32[0], h64[0] if h64 else None, 4 + k, got3))
    return runs


C2_SLICE = 5        # the slice of the batch of eight that the CPU oracle replays


def _config_c2_submit(pool):
    """BASELINE config C2 at its literal setting: a batch of EIGHT 512x512 low-dose images (global slice ids 0..7), image domain
    only, t_start_img=[15], constant guidance, no ultra pass, the production image UNet -- through the drop-in
    img_denoiser(mode="img_only") (Utils/train_test_utils.py:482-550)."""
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, _RecordingNoise
    from ipdm_pytorch_amd.diffusion import NoiseSource
    from tests import _oracle_child as oc
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(device=DEV, mode="test_img", t_start_img=[15], ultra_img_denoise=False), opt.__dict__)
    x = np.stack([synth.rasterize(synth.ellipse_phantom(b)) + 0.004 * synth.hash_normal((512, 512), 500 + b) for b in range(8)]).astype(np.float32)
    den = progressive_domain_denoiser(opt, seed=77)
    den.noise = _RecordingNoise(NoiseSource(77, 0))
    got = den.img_denoiser(torch.from_numpy(x)[:, None], noise_strength=None, mode="img_only").cpu().numpy()
    b = C2_SLICE
    job = pool.path("c2_slice%d.npz" % b)
    oc.write_job(job, opt.__dict__, x[b], [z[b:b + 1].cpu().numpy() for z in den.noise.draws], 0, 70, img_only=True)
    h = pool.submit("C2 slice %d f32" % b, job, 3)
    # a batch is its slices: the replayed slice and one more, each sampled alone on the device (global slice id kept), bit for bit
    alone = {}
    for s_id in (b, 0):
        one = progressive_domain_denoiser(opt, seed=77, slice_id0=s_id)
        alone[s_id] = one.img_denoiser(torch.from_numpy(x[s_id:s_id + 1])[:, None], noise_strength=None, mode="img_only").cpu().numpy()
        del one
    del den
    torch.cuda.empty_cache()
    return dict(got=got, h=h, alone=alone)


# ---- device halves, in the order the pool should start their replays: the longest first
@pytest.mark.oracle_submit
def test_headline_configuration_device_run(oracle_pool):
    """Device half of test_headline_configuration_full_length: B = 2 at the benched length + slice 0 alone; slice 1's replay
    (75 network evaluations on nine of the box's sixteen CPUs, ~7 min) starts here, at the head of the session."""
    st = _once(oracle_pool, "headline", _headline_submit)
    assert st["got"].shape == (2, 1, 512, 512) and np.isfinite(st["got"]).all()
    assert np.array_equal(st["alone"], st["got"][0:1]), float(np.abs(st["alone"] - st["got"][0:1]).max())     # a batch is its slices


@pytest.mark.oracle_submit
def test_full_size_pipeline_device_run(oracle_pool):
    """Device half of test_full_size_pipeline_psnr (FULL_SIZE_SEEDS; float32 replays, float64 ones for the first FULL_SIZE_F64)."""
    runs = _once(oracle_pool, "full_size", _full_size_submit)
    assert len(runs) == len(FULL_SIZE_SEEDS) and all(np.isfinite(r[0]).all() for r in runs)


@pytest.mark.oracle_submit
def test_config_c2_device_run(oracle_pool):
    """Device half of test_config_c2_batch_of_eight_img_only."""
    st = _once(oracle_pool, "c2", _config_c2_submit)
    assert st["got"].shape == (8, 1, 512, 512) and np.isfinite(st["got"]).all()
    for s_id, a in st["alone"].items():
        assert np.array_equal(a, st["got"][s_id:s_id + 1]), (s_id, float(np.abs(a - st["got"][s_id:s_id + 1]).max()))


@pytest.mark.oracle_submit
def test_smoke_pipeline_device_runs(oracle_pool):
    """Device half of the three reduced-pipeline tests (PSNR of the canonical seed, stage-by-stage arbiter, arbiter over the seeds)."""
    runs = _once(oracle_pool, "reduced", _reduced_submit)
    assert len(runs) == len(REDUCED_SEEDS) and all(r["out"].shape == (1, 1, 512, 512) and np.isfinite(r["out"]).all() for r in runs)


@pytest.mark.oracle_join
def test_config_c2_batch_of_eight_img_only(oracle_pool):
    """BASELINE.json config C2 ("Batch=8 512x512 slices, image-domain UNet only, t_start_img=[15]") against the CPU oracle: the
    production image UNet, 15 network evaluations per slice, no ultra pass.  Slice C2_SLICE of the batch is replayed by the
    oracle with the device's draws (max-abs 1e-4 relative, PSNR within 1e-4 relative: north_star); that slice and slice 0 equal
    their single-slice runs bit for bit (checked by the device half), so every slice is what the reference's one-slice-at-a-time
    call (Utils/train_test_utils.py:290-294) defines."""
    st = _once(oracle_pool, "c2", _config_c2_submit)
    want = oracle_pool.result(st["h"])
    rep = _check_full_size(st["got"][C2_SLICE:C2_SLICE + 1], want, C2_SLICE, FULL_SIZE_MAX_REL)
    print("config C2 B=8 img-only t_start_img=[15]: slice %d vs CPU oracle max-abs %.3e rms %.3e PSNR hip %.4f / cpu %.4f dB" % ((C2_SLICE,) + rep))


@pytest.mark.oracle_join
def test_full_size_pipeline_psnr(oracle_pool):
    """End to end at full size with the production architectures, few steps, over FULL_SIZE_SEEDS (three by default, FIVE
    under IPDM_PARITY_FULL=1; weights fixed; phantom, dose noise and diffusion draws vary): proj loop with adaptive guidance ->
    FBP -> sharpen -> img loop -- against the float32 CPU oracle (north_star's PSNR criterion, max-abs 1e-4), and, for the
    first FULL_SIZE_F64 seeds (one by default, all under IPDM_PARITY_FULL=1) replayed once more in FLOAT64, the fp64 arbiter on
    the PRODUCTION kernels (conv_wino2 / conv_wino / conv_ws / conv_direct / attention_ws / the parity form): err(HIP, fp64) /
    err(oracle32, fp64) under the hard caps for every seed and, with five seeds or more, its median <= 1.25 in rms and <= 1.5
    in max-abs (nothing amplifies at full size: round 4-5 measured 0.99 ... 1.03 rms, 0.93 ... 1.24 max-abs over five seeds)."""
    runs = _once(oracle_pool, "full_size", _full_size_submit)
    wants = [oracle_pool.result(r[1]) for r in runs]
    report = [_check_full_size(got, want, ph, FULL_SIZE_MAX_REL) for (got, _, _, ph, _), want in zip(runs, wants)]
    print("full-size %d seeds: max-abs %s rms %s" % (len(runs), ["%.2e" % r[0] for r in report], ["%.2e" % r[1] for r in report]))
    arb = [(r[0], w, oracle_pool.result(r[2])) for r, w in zip(runs, wants) if r[2] is not None]
    _arbiter_ratios("full size", [a[0] for a in arb], [a[1] for a in arb], [a[2] for a in arb])


@pytest.mark.oracle_join
def test_full_size_pipeline_bf16x3_alt_mode(oracle_pool):
    """The OPT-IN evaluation conv_bf16x3 (conv_wino3.hip: the wide 3x3 layers' products on the bf16 matrix pipe through an
    error-free three-way split, float32 accumulate) through the gates that define parity here: the full-size samples of
    test_full_size_pipeline_psnr once more with the option on -- same inputs, weights and draws -- against the SAME CPU oracle
    replays: max-abs 1e-4 relative, PSNR within 1e-4 relative (north_star), and against the float64 replays err(HIP, fp64) /
    err(oracle32, fp64) under the arbiter's caps (the median criterion with five seeds: IPDM_PARITY_FULL=1)."""
    runs = _once(oracle_pool, "full_size", _full_size_submit)
    wants = [oracle_pool.result(r[1]) for r in runs]
    report = [_check_full_size(r[4], want, r[3], FULL_SIZE_MAX_REL) for r, want in zip(runs, wants)]
    print("full-size %d seeds, conv_bf16x3: max-abs %s rms %s; against the default path max-abs %s" % (
        len(runs), ["%.2e" % r[0] for r in report], ["%.2e" % r[1] for r in report], ["%.2e" % float(np.abs(r[4] - r[0]).max()) for r in runs]))
    assert all(not np.array_equal(r[4], r[0]) for r in runs)          # (the option did route layers to the other kernel)
    arb = [(r[4], w, oracle_pool.result(r[2])) for r, w in zip(runs, wants) if r[2] is not None]
    _arbiter_ratios("full size conv_bf16x3", [a[0] for a in arb], [a[1] for a in arb], [a[2] for a in arb])


@pytest.mark.run_last
def test_bf16x3_first_forward_of_a_process():
    """The defect that kept conv_bf16x3's faster wave order out of the library (conv_wino3.hip, IPDM_WINO3_STAGGER; NOTEBOOK.md round 6): with any
    wave multiplying before it staged, the FIRST forward of a process was wrong (1e-2 relative in one workgroup tile of one convolution) in 30 - 50 %
    of fresh processes on two of the boxes seen -- later forwards, blocking launches and every in-process repetition test were clean, so nothing in
    the suite saw it until the whole suite ran under the option.  Four fresh processes of the shipped order (and one of the float32 kernels):
    every forward bit-equal to the process's fourth.  The instruction behind it is identified (a packed add of the input transform whose low
    result takes the high half of a source: `a.lo + 0` instead of `a.lo - b.hi` in lanes 48-63) and replaced, and the shipped order never
    showed it (0 of 60 fresh processes); WHY that instruction fails there is not known, so the check stays -- collected LAST (marker
    run_last): should it fail on some box, `pytest -x` has run everything else by then."""
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_first_forward_child.py")
    for on in (1, 1, 1, 1, 0):
        r = subprocess.run([sys.executable, child, str(on)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (on, r.returncode, r.stdout[-300:], r.stderr[-600:])


def test_bf16x3_pipeline_batch_is_its_slices():
    """Option conv_bf16x3 through the whole path at full size (production UNets, 2000x912 sinograms, adaptive guidance, FBP,
    sharpen, image domain, ultra pass; two steps per stage): a batch of THREE slices equals, bit for bit, each of its slices
    sampled alone (global slice id kept) -- the property sharding across ranks rests on, which the default path has because its
    batch-dependent kernel choices are between bit-identical kernels, and which the option keeps by choosing conv_wino3 by the
    layer alone."""
    from ipdm_pytorch_amd import _lib
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(t_start_proj=[2, 2], t_start_img=[2], ultra_img_denoise=True, device=DEV), opt.__dict__)
    sinos = np.stack([synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(p)), seed=p) for p in (0, 1, 2)])
    with _lib.option("conv_bf16x3", 1):
        den = progressive_domain_denoiser(opt, seed=321)
        den.data_sample_load(ldproj=torch.from_numpy(sinos)[:, None])
        whole = den.progressive_denoiser(sharpen_num=70).cpu().numpy()
        del den
        for b in (0, 2):
            one = progressive_domain_denoiser(opt, seed=321, slice_id0=b)
            one.data_sample_load(ldproj=torch.from_numpy(sinos[b:b + 1])[:, None])
            alone = one.progressive_denoiser(sharpen_num=70).cpu().numpy()
            del one
            assert np.array_equal(alone, whole[b:b + 1]), (b, float(np.abs(alone - whole[b:b + 1]).max()))
    den = progressive_domain_denoiser(opt, seed=321)
    den.data_sample_load(ldproj=torch.from_numpy(sinos)[:, None])
    ref = den.progressive_denoiser(sharpen_num=70).cpu().numpy()
    del den
    torch.cuda.empty_cache()
    d = float(np.abs(ref - whole).max())
    print("conv_bf16x3, B = 3 full size: batch == slices (bitwise); against the default path max-abs %.2e" % d)
    assert 0.0 < d <= FULL_SIZE_MAX_REL * max(1.0, float(np.abs(ref).max()))


@pytest.mark.oracle_join
def test_headline_configuration_full_length(oracle_pool):
    """The BENCHED configuration at its real length (Utils/train_test_utils.py:552-567, Model/model.py:517-642):
    production UNets, 2000x912 sinograms, t_start_proj=[15,15,15] (adaptive guidance), FBP, sharpen, t_start_img=[15],
    ultra pass = 45 proj + 30 img network evaluations per slice, as a batch of TWO slices (global ids 0, 1) on the device.
    Slice 1 -- the one whose batch index, noise key and statistics rows are not those of a batch of one -- is replayed by
    the CPU oracle with the recorded draws (a pinned child process, ~5 min of CPU, started by the device half at the head of
    the session); slice 0 must equal the same slice sampled alone on the device, bit for bit (a batch is its slices), and
    B = 1 runs are what test_full_size_pipeline_psnr holds against the oracle.  (All eight slices of the BENCHED batch against
    the oracle: tools/b8_vs_oracle.py, profiles/r06_b8_vs_oracle.txt.)"""
    st = _once(oracle_pool, "headline", _headline_submit)
    got, alone = st["got"], st["alone"]
    want = oracle_pool.result(st["h"], timeout=1100.0)
    rep = _check_full_size(got[1:2], want, 1, FULL_SIZE_MAX_REL)
    assert np.array_equal(alone, got[0:1]), float(np.abs(alone - got[0:1]).max())
    msg = "headline full length B=2: slice 1 vs CPU oracle max-abs %.3e rms %.3e PSNR hip %.4f / cpu %.4f dB; slice 0 == the slice alone (bitwise)" % rep
    print(msg)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "headline_parity.txt"), "w") as f:
        f.write(msg + "\n")


def test_result_dicts_with_saved_states():
    """save_it_state_proj / save_it_state_img / save_proj_state=True: every stored iterate (ResultTempDict 'iter_k',
    Utils/train_test_utils.py:459-479,541-548) against the oracle's intermediates, incl. the FBP of EVERY proj iterate."""
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, SMOKE_PROJ, SMOKE_IMG, _RecordingNoise
    from ipdm_pytorch_amd.diffusion import NoiseSource
    from ipdm_pytorch_amd.unet import UNetModel
    from oracle import pipeline as op, fbp as of
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(device=DEV, t_start_proj=[2, 2], t_start_img=[2, 1], ultra_img_denoise=False, save_it_state_proj=True,
                  save_it_state_img=True), opt.__dict__)
    den = progressive_domain_denoiser(opt, seed=9)
    den.proj_model = UNetModel(**SMOKE_PROJ).to(DEV)
    den.img_model = UNetModel(**SMOKE_IMG).to(DEV)
    sd_p = synth.synth_state_dict(den.proj_model._shapes, seed=21)
    sd_i = synth.synth_state_dict(den.img_model._shapes, seed=22)
    den.proj_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_p.items()})
    den.img_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_i.items()})
    sino = synth.low_dose(synth.fan_sinogram(synth.ellipse_phantom(5)), seed=5)
    den.data_sample_load(ldproj=torch.from_numpy(sino)[None, None])
    rec = _RecordingNoise(NoiseSource(9, 0))
    den.noise = rec
    out = den.progressive_denoiser(save_proj_state=True, sharpen_num=42)
    cfg_p = ou.UNetConfig(1, 16, 1, attention_resolutions=(16,), channel_mult=(0.25, 0.25, 0.5, 1, 2, 4), num_heads=1)
    cfg_i = ou.UNetConfig(1, 16, 1, attention_resolutions=(8,), channel_mult=(1, 1, 2, 2, 4), num_heads=1)
    draws = iter([z.cpu() for z in rec.draws])
    want, mid = op.progressive_slice(dict(opt.__dict__), cfg_p, {k: torch.from_numpy(v) for k, v in sd_p.items()}, cfg_i,
                                     {k: torch.from_numpy(v) for k, v in sd_i.items()}, torch.from_numpy(sino)[None, None],
                                     lambda: next(draws), sharpen_num=42)
    # proj iterates: 2 passes + their mean = 3 entries, each also converted by FBP
    assert len(den.proj_denoise_result) == len(mid["proj"]) == 3
    assert len(den.proj_denoise_convert2img_result) == 3
    geo = of.FBPGeometry()
    for k in range(3):
        np.testing.assert_allclose(den.proj_denoise_result[k + 1], mid["proj"][k].numpy(), rtol=0, atol=2e-4)
        fbp_k = of.convert(geo, mid["proj"][k][:, 0].numpy())[:, None]
        np.testing.assert_allclose(den.proj_denoise_convert2img_result[k + 1], fbp_k, rtol=0, atol=2e-4 * max(1.0, np.abs(fbp_k).max()))
    # img iterates: 2 passes + mean
    assert len(den.progressive_denoise_result) == len(mid["img"]) == 3
    for k in range(3):
        np.testing.assert_allclose(den.progressive_denoise_result[k + 1], mid["img"][k].numpy(), rtol=0, atol=3e-4)
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=0, atol=3e-4)
    assert den.progressive_denoise_result[-1] is den.progressive_denoise_result["iter_3"]
