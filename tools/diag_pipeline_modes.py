#!/usr/bin/env python
"""The benched sample (B slices, full length unless shortened) under the per-call kernel switches against the default
path with the SAME noise draws: max-abs / rms / PSNR of the final images.  Every switch is another float32 evaluation of
the same function; the sample amplifies rounding, so expect 1e-5 .. 1e-4 of the output range, never structure.
    python tools/diag_pipeline_modes.py [B] [t_proj] [t_img] [step]     (step k: the k-th consecutive sample of the noise stream, as bench.py --steps k)"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
import bench                                            # noqa: E402
from ipdm_pytorch_amd import _lib                       # noqa: E402
from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options      # noqa: E402
from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser                 # noqa: E402
from ipdm_pytorch_amd.diffusion import NoiseSource      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tp = int(sys.argv[2]) if len(sys.argv) > 2 else 15
ti = int(sys.argv[3]) if len(sys.argv) > 3 else 15
STEP = int(sys.argv[4]) if len(sys.argv) > 4 else 1
device = "cuda:0"
opt = default_cfg([])
cfg_load(mayo_test_options(), opt.__dict__)
cfg_load(dict(device=device, t_start_proj=[tp, tp, tp], t_start_img=[ti], ultra_img_denoise=True), opt.__dict__)
ldproj = bench.make_inputs(B, 0, device, None)


def run():
    den = progressive_domain_denoiser(opt, seed=1234, slice_id0=0)
    den.data_sample_load(ldproj=ldproj)
    for _ in range(STEP):
        out = den.progressive_denoiser_device(sharpen_num=70)
    torch.cuda.synchronize()
    return out.float().cpu()


base = run()
again = run()
print("default twice: bitwise equal =", torch.equal(base, again), flush=True)
rng = float(base.max() - base.min())
MODES = (("conv_no_up2", 1), ("direct_no_skip_fuse", 1), ("conv_no_wino", 1), ("gn_unfused", 1), ("conv1x1_no_quarter", 1))
if os.environ.get("MODES"):
    MODES = tuple((m, 1) for m in os.environ["MODES"].split(","))
for name, val in MODES:
    with _lib.option(name, val):
        o = run()
    d = o - base
    per = d.abs().flatten(1).max(1).values
    mse = float((d * d).mean())
    print("%-22s max %.2e rms %.2e PSNR %.1f dB (range %.3f) per-slice max %s" % (
        name, float(d.abs().max()), math.sqrt(mse), 10 * math.log10(rng * rng / mse) if mse > 0 else float("inf"), rng,
        " ".join("%.1e" % float(v) for v in per)), flush=True)
    if float(d.abs().max()) > 1e-4:      # where: a structured region (a guidance-map block flipping a threshold) or scattered?
        sl = int(per.argmax())
        m = d[sl, 0].abs() > 0.25 * float(d.abs().max())
        ys, xs = m.nonzero(as_tuple=True)
        print("    slice %d: %d pixels above a quarter of the maximum, rows %d..%d, columns %d..%d; >1e-4: %d pixels" % (
            sl, int(m.sum()), int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max()), int((d[sl, 0].abs() > 1e-4).sum())), flush=True)
