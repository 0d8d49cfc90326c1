"""Oracle self-check against the IMPORTED reference (build container only; /root/reference is not on
the GPU box and nothing under tests -m gpu / smoke() / bench.py imports this file).

TEST INFRASTRUCTURE ONLY.  Complements tests/golden/ (small fixtures) with FULL-SIZE comparisons that
are too big to commit:
  1. both reference UNet configurations at their real resolution (img 512x512, proj 2000x912), the
     oracle's functional restatement vs the reference nn.Module on the same seeded state_dict;
  2. guided_reverse_process (constant and adaptive guidance, img and proj mode) on a reduced UNet at a
     mid-size image, reference loop (B=1) vs the oracle's per-slice loop with the same injected noise;
  3. every FBP geometry table, complete, and ramp-filtered rows.

Run:  python oracle/check_vs_reference.py [--quick]
Prints one line per check and exits non-zero on any failure.  Last run is recorded in DESIGN.md.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, unet as ou, diffusion as od, fbp as of  # noqa: E402
import ipdm_pytorch_amd  # noqa: E402,F401
from ipdm_pytorch_amd import synth  # noqa: E402

FAILED = []


def report(name, err, tol, extra=""):
    ok = err <= tol
    print("%-58s err %.3e  tol %.1e  %s %s" % (name, err, tol, "ok" if ok else "FAIL", extra))
    if not ok:
        FAILED.append(name)


def rel_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def check_unets(M, quick):
    torch.set_num_threads(os.cpu_count() or 1)
    cases = {
        "img": (dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=[8, 16],
                     channel_mult=[1, 1, 2, 2, 4, 4]), ou.UNetConfig(), (1, 1, 512, 512)),
        "proj": (dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=[16, 32],
                      channel_mult=[1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4]),
                 ou.UNetConfig(attention_resolutions=(16, 32), channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4)),
                 (1, 1, 2000, 912)),
    }
    for name, (kw, cfg, shape) in cases.items():
        if quick:
            shape = (1, 1, shape[2] // 4, shape[3] // 4)
        net = M.UNetModel(**kw).eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert shapes == ou.param_shapes(cfg), "state_dict layout of the oracle differs from the reference (%s)" % name
        sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=5).items()}
        net.load_state_dict(sd)
        x = torch.from_numpy(synth.hash_normal(shape, 6))
        for t in (0, 14):
            t0 = time.perf_counter()
            with torch.no_grad():
                want = net(x, torch.full((1,), t, dtype=torch.long)).numpy()
            t1 = time.perf_counter()
            got = ou.unet_forward(cfg, sd, x, t).numpy()
            t2 = time.perf_counter()
            report("unet %s %s t=%d" % (name, "x".join(map(str, shape[2:])), t), rel_max(got, want), 2e-5,
                   "(ref %.1fs oracle %.1fs)" % (t1 - t0, t2 - t1))


class _Feed:
    def __init__(self, seed):
        self.seed, self.k = seed, 0

    def __call__(self, x, *a, **k):
        z = torch.from_numpy(synth.hash_normal(tuple(x.shape), self.seed * 1000 + self.k))
        self.k += 1
        return z


def check_loops(M, U):
    kw = dict(in_channels=1, model_channels=16, out_channels=1, attention_resolutions=[4], channel_mult=[1, 1, 2, 2],
              num_heads=1)
    cfg = ou.UNetConfig(1, 16, 1, attention_resolutions=(4,), channel_mult=(1, 1, 2, 2), num_heads=1)
    net = M.UNetModel(**kw).eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, seed=9).items()}
    net.load_state_dict(sd)
    curves = {"img": U.curve_init(), "proj": U.proj_curv_init()}
    cases = {
        "img const 0.45 [4,3] clip": ("img", (1, 1, 96, 96), 1, dict(t_start=[4, 3], clip=True, lambda_ratio=10, eta=0.7,
                                                                    constant_guidance=0.45)),
        "img adaptive-guidance [3,3,2]": ("img", (1, 1, 96, 96), 1, dict(t_start=[3, 3, 2], clip=True, lambda_ratio=10,
                                                                        eta=0.7, constant_guidance=None)),
        "proj adaptive-guidance [3,3,3]": ("proj", (1, 1, 200, 96), 5, dict(t_start=[3, 3, 3], clip=False, lambda_ratio=1,
                                                                           eta=0.5, constant_guidance=None)),
        "proj t_start=None (adaptive passes)": ("proj", (1, 1, 120, 64), 5, dict(t_start=None, clip=False,
                                                                                lambda_ratio=1, eta=0.5,
                                                                                constant_guidance=None)),
    }
    orig = torch.randn_like
    try:
        for tag, (mode, shape, power, k) in cases.items():
            gd = M.GaussianDiffusion(timesteps=1000, beta_schedule="cosine", schedule_power=power)
            img = torch.from_numpy(synth.hash_uniform(shape, 42)) * (0.05 if mode == "img" else 0.6) + (0.17 if mode == "img" else 0)
            ldct = torch.from_numpy(synth.hash_uniform(shape, 44)) * 0.05 + 0.17
            torch.randn_like = _Feed(45)
            res, _, ns = gd.guided_reverse_process(
                model=net, img=img, mode=mode, save_states=False, lambda_curve=curves[mode], ldct=ldct,
                kernel_size_img=4, amplitude_img=30, kernel_size_proj=4, amplitude_proj=7, only_convertor=False,
                normal=False, noise_strength=None, transformer=None, **k)
            torch.randn_like = orig
            feed = _Feed(45)
            got, ns2 = od.guided_reverse_process_slice(
                od.Schedule(1000, power), lambda x, t: ou.unet_forward(cfg, sd, x, t), img, t_start=k["t_start"],
                clip=k["clip"], lambda_ratio=k["lambda_ratio"], eta=k["eta"], mode=mode,
                constant_guidance=k["constant_guidance"], noise_fn=lambda: feed(img), ldct=ldct, kernel_size=4,
                amplitude=30 if mode == "img" else 7)
            assert len(got) == len(res), "%s: %d iterates vs %d" % (tag, len(got), len(res))
            assert ns2 == ns, "%s: noise_strength %r vs %r" % (tag, ns2, ns)
            err = max(rel_max(g.numpy(), r.numpy()) for g, r in zip(got, res))
            report("loop %s" % tag, err, 5e-5, "(%d iterates, %d draws)" % (len(res), feed.k))
    finally:
        torch.randn_like = orig


def check_fbp(FB):
    ref = FB.FBP("cpu")
    geo = of.FBPGeometry()
    for name, got, want in (("theta", geo.theta, ref.theta), ("nda", geo.nda, ref.nda), ("h_RL", geo.h_RL, ref.h_RL),
                            ("r", geo.r, ref.r), ("phi", geo.phi, ref.phi)):
        same = np.asarray(got).dtype == np.asarray(want).dtype and np.array_equal(got, want)
        report("fbp table %s (complete, bitwise, dtype %s)" % (name, np.asarray(want).dtype), float(not same), 0.0)
    rows = (synth.hash_uniform((1, 16, 912), 51) * 4.0).astype(np.float32)
    want = FB.conv_pj(np.zeros_like(rows), rows, ref.h_RL, 16, 912, 1)
    g16 = of.FBPGeometry(n_views=16)
    report("fbp ramp rows vs conv_pj (np.convolve)", rel_max(of.ramp_filter(g16, rows), want), 2e-6)


def main():
    quick = "--quick" in sys.argv
    M, FB = ref_shim.load("numpy")      # full-size maps: the vectorised restatement (make_golden pins it to the py_func)
    U = ref_shim.load_curves()
    check_fbp(FB)
    check_loops(M, U)
    check_unets(M, quick)
    if FAILED:
        print("FAILED: %s" % FAILED)
        raise SystemExit(1)
    print("oracle agrees with the imported reference on every check")


if __name__ == "__main__":
    main()
