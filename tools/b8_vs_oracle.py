#!/usr/bin/env python
"""The BENCHED batch against the CPU oracle, slice by slice (Utils/train_test_utils.py:290-294 calls the
sampler one slice at a time, so per-slice semantics are the contract -- shown here on the batch that bench.py times).

Device: bench.py's default workload -- B = 8 synthetic 0.25-dose slices (global ids 0..7), seed 1234, production UNets,
t_start_proj=[15,15,15], t_start_img=[15], ultra pass -- with the device's draws recorded.  Host: eight pinned CPU oracle
replays side by side (tests/_oracle_child.py through tests/_oracle_pool.py: test infrastructure, the checker only), one per
slice.  Report: per-slice max-abs, rms and the PSNR pair (vs the phantom, on miu2pixel images) -> gpurun_out/<tag>_b8_vs_oracle.txt.

    python tools/b8_vs_oracle.py [tag] [threads per replay]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 2       # (8 replays x 2 threads = the 16 CPUs the boxes' cgroup grants)
    import numpy as np
    import torch
    import bench
    from ipdm_pytorch_amd import synth
    from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
    from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser, _RecordingNoise
    from ipdm_pytorch_amd.diffusion import NoiseSource
    from oracle import diffusion as od
    from tests import _oracle_child as oc
    from tests._oracle_pool import OraclePool
    B, dev = 8, "cuda:0"
    opt = default_cfg([])
    cfg_load(mayo_test_options(), opt.__dict__)
    cfg_load(dict(device=dev, t_start_proj=[15, 15, 15], t_start_img=[15], ultra_img_denoise=True), opt.__dict__)
    den = progressive_domain_denoiser(opt, seed=1234, slice_id0=0)
    ldproj = bench.make_inputs(B, 0, dev)                     # exactly the bench's inputs
    den.data_sample_load(ldproj=ldproj)
    rec = _RecordingNoise(NoiseSource(1234, 0))
    den.noise = rec
    t0 = time.time()
    got = den.progressive_denoiser_device(sharpen_num=70)
    torch.cuda.synchronize()
    t_dev = time.time() - t0
    got = got.cpu().numpy()
    sinos = ldproj[:, 0].cpu().numpy()
    pool = OraclePool(reserve_main=0)                         # (this process only waits)
    hs = []
    for b in range(B):
        job = pool.path("b8_slice%d.npz" % b)
        oc.write_job(job, opt.__dict__, sinos[b], [z[b:b + 1].cpu().numpy() for z in rec.draws], 0, 70)
        hs.append(pool.submit("b8 slice %d" % b, job, threads))
    del rec, den
    torch.cuda.empty_cache()
    lines = ["B = 8 benched batch (seed 1234, global slice ids 0..7, t_start_proj=[15,15,15], t_start_img=[15], ultra) against the CPU oracle, "
             "slice by slice; device pass %.2f s (first pass of the process, draws recorded); %d replay threads per slice" % (t_dev, threads)]
    worst = [0.0, 0.0]
    ok = True
    for b in range(B):
        want = pool.result(hs[b], timeout=3000.0)
        err = np.abs(got[b:b + 1].astype(np.float64) - want)
        scale = max(1.0, float(np.abs(want).max()))
        truth = od.miu2pixel(torch.from_numpy(synth.rasterize(synth.ellipse_phantom(b % 16)))).numpy()
        p_hip = od.psnr(truth, od.miu2pixel(torch.from_numpy(got[b, 0])).numpy())
        p_cpu = od.psnr(truth, od.miu2pixel(torch.from_numpy(want[0, 0])).numpy())
        good = err.max() <= 1e-4 * scale and abs(p_hip - p_cpu) <= 1e-4 * abs(p_cpu)
        ok = ok and good
        worst = [max(worst[0], float(err.max())), max(worst[1], abs(p_hip - p_cpu) / abs(p_cpu))]
        lines.append("slice %d: max-abs %.3e rms %.3e (scale %.3f) | PSNR hip %.5f dB / oracle %.5f dB (rel %.1e) | %s" % (
            b, err.max(), float(np.sqrt((err ** 2).mean())), scale, p_hip, p_cpu, abs(p_hip - p_cpu) / abs(p_cpu), "ok" if good else "OUT OF BOUND"))
        big = np.argwhere(err[0, 0] > 5e-5 * scale)
        if len(big):      # where: a localized streak (a guidance-map block across the jump of weight_lambda) or spread over the image?
            lines.append("         %d pixels above 5e-5: rows %d..%d, columns %d..%d; the 99.9th percentile of |err| is %.2e" % (
                len(big), big[:, 0].min(), big[:, 0].max(), big[:, 1].min(), big[:, 1].max(), float(np.quantile(err, 0.999))))
    lines.append("worst: max-abs %.3e (bound 1e-4 x scale), PSNR relative difference %.1e (bound 1e-4): %s" % (worst[0], worst[1], "ALL WITHIN BOUNDS" if ok else "FAILED"))
    lines.append(pool.report())
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "%s_b8_vs_oracle.txt" % tag), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    pool.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
