"""CPU child process of the full-size GPU parity tests: replays ONE slice of the dual-domain sample through the CPU oracle
(oracle.pipeline.progressive_slice) with the draws the device recorded, and saves the result.  Several of these run side
by side (one per slice / seed) so that a 75-forward oracle run fits the GPU test budget.  Never touches the GPU.

  python tests/_oracle_child.py job.npz out.npy n_threads [cpu,cpu,...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL_IMG = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(8, 16), channel_mult=(1, 1, 2, 2, 4, 4))
FULL_PROJ = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(16, 32),
                 channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4))
OPT_KEYS = ("timesteps_proj", "schedule_power_proj", "t_start_proj", "clip_proj", "lambda_ratio_proj", "eta_proj",
            "constant_guidance_proj", "kernel_size_proj", "amplitude_proj", "timesteps_img", "schedule_power_img", "t_start_img",
            "clip_img", "lambda_ratio_img", "eta_img", "constant_guidance_img", "kernel_size_img", "amplitude_img", "convertor",
            "fbp_sharpen", "ultra_img_denoise")


SMOKE_PROJ = dict(in_channels=1, model_channels=16, out_channels=1, attention_resolutions=(16,), channel_mult=(0.25, 0.25, 0.5, 1, 2, 4), num_heads=1)
SMOKE_IMG = dict(in_channels=1, model_channels=16, out_channels=1, attention_resolutions=(8,), channel_mult=(1, 1, 2, 2, 4), num_heads=1)


def write_job(path, opt_dict, sino, draws, weight_seed, sharpen_num, dtype="float32", nets="full", mid=False, img_only=False):
    """draws: the recorded [1,1,h,w] tensors/arrays of ONE slice, in order (sinogram-shaped ones first, then image-shaped).
    dtype "float64": the ARBITER run -- the same float32 inputs, weights and draws evaluated in double precision.
    nets "full": the production architectures, both with synthetic weights of seed `weight_seed`; "smoke": the reduced
    16-channel networks of ipdm_pytorch_amd.denoiser.SMOKE_PROJ / SMOKE_IMG with weight seeds (weight_seed, weight_seed + 1).
    mid: the child saves every stored iterate too (out.npy -> out.npy + out.npy.mid.npz: proj [n,h,w], fbp [G,G], img [m,G,G]).
    img_only: `sino` is a [G,G] IMAGE and the child runs the image-domain half alone -- img_denoiser(mode="img_only") of the
    reference harness (Utils/train_test_utils.py:482-550: the img loop, then the ultra loop if the option asks for it)."""
    import numpy as np
    draws = [np.asarray(d, dtype=np.float32).reshape(d.shape[-2], d.shape[-1]) for d in draws]
    n_p = 0 if img_only else sum(1 for d in draws if d.shape == tuple(sino.shape))
    assert img_only or (all(d.shape == tuple(sino.shape) for d in draws[:n_p]) and all(d.shape != tuple(sino.shape) for d in draws[n_p:]))
    np.savez(path, sino=np.asarray(sino, dtype=np.float32), draws_p=np.stack(draws[:n_p]) if n_p else np.zeros((0,) + tuple(sino.shape), np.float32),
             draws_i=np.stack(draws[n_p:]), img_only=int(bool(img_only)), opt=json.dumps({k: opt_dict[k] for k in OPT_KEYS}), weight_seed=weight_seed, sharpen_num=sharpen_num, dtype=dtype, nets=nets, mid=int(bool(mid)))


def cpu_blocks(njobs, threads):
    """Disjoint blocks of `threads` logical CPUs for njobs side-by-side children, spread over the first half of the CPU
    list (on the GPU boxes -- 2 x 64 cores x 2 SMT -- the physical cores of both sockets; SMT siblings are the second half):
    unpinned, five 32-thread torch processes ran 7x slower than one alone (memory-bandwidth and thread migration)."""
    ncpu = os.cpu_count() or 8
    phys = ncpu // 2 if ncpu >= 16 else ncpu
    threads = max(1, min(threads, phys // max(1, njobs)))
    stride = phys // njobs
    return threads, [list(range(i * stride, i * stride + threads)) for i in range(njobs)]


def run_jobs(jobs, threads):
    """jobs: [(job.npz, out.npy)]; runs them as parallel, CPU-pinned child processes, returns the outputs."""
    import subprocess
    import numpy as np
    threads, blocks = cpu_blocks(len(jobs), threads)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), j, o, str(threads), ",".join(map(str, blk))], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for (j, o), blk in zip(jobs, blocks)]
    outs = []
    for p, (j, o) in zip(procs, jobs):
        log, _ = p.communicate(timeout=1500)
        assert p.returncode == 0, log[-3000:]
        outs.append(np.load(o))
    return outs


def main():
    job, out, threads = sys.argv[1], sys.argv[2], int(sys.argv[3])
    if len(sys.argv) > 4 and sys.argv[4]:
        try:
            os.sched_setaffinity(0, {int(c) for c in sys.argv[4].split(",")})
        except (OSError, ValueError):
            pass
    import numpy as np
    import torch
    torch.set_num_threads(threads)
    from ipdm_pytorch_amd import synth
    from oracle import pipeline as op, unet as ou
    j = np.load(job, allow_pickle=False)
    opt = json.loads(str(j["opt"]))
    smoke = "nets" in j.files and str(j["nets"]) == "smoke"
    dt = torch.float64 if ("dtype" in j.files and str(j["dtype"]) == "float64") else torch.float32
    cfg_p, cfg_i = (ou.UNetConfig(**SMOKE_PROJ), ou.UNetConfig(**SMOKE_IMG)) if smoke else (ou.UNetConfig(**FULL_PROJ), ou.UNetConfig(**FULL_IMG))
    seed = int(j["weight_seed"])
    sd_p = {k: torch.from_numpy(v).to(dt) for k, v in synth.synth_state_dict(ou.param_shapes(cfg_p), seed=seed).items()}
    sd_i = {k: torch.from_numpy(v).to(dt) for k, v in synth.synth_state_dict(ou.param_shapes(cfg_i), seed=seed + (1 if smoke else 0)).items()}
    draws = iter([torch.from_numpy(d)[None, None].to(dt) for d in j["draws_p"]] + [torch.from_numpy(d)[None, None].to(dt) for d in j["draws_i"]])
    if "img_only" in j.files and int(j["img_only"]):
        from oracle import diffusion as od
        x = torch.from_numpy(j["sino"])[None, None].to(dt)
        sch = od.Schedule(opt["timesteps_img"], opt["schedule_power_img"])
        eps = lambda xx, t: ou.unet_forward(cfg_i, sd_i, xx, t)   # noqa: E731
        kw = dict(clip=opt["clip_img"], lambda_ratio=opt["lambda_ratio_img"], mode="img", noise_fn=lambda: next(draws), ldct=x,
                  kernel_size=opt["kernel_size_img"], amplitude=opt["amplitude_img"], noise_strength_in=None)
        res, _ = od.guided_reverse_process_slice(sch, eps, x, t_start=opt["t_start_img"], eta=opt["eta_img"],
                                                 constant_guidance=opt["constant_guidance_img"], **kw)
        if opt["ultra_img_denoise"]:
            res_u, _ = od.guided_reverse_process_slice(sch, eps, res[-1], t_start=[5, 5, 5], eta=0.6, constant_guidance=0.6, **kw)
            res = res + res_u
        assert next(draws, None) is None, "the oracle consumed fewer draws than the device recorded"
        np.save(out, res[-1].numpy())
        return
    want, mid = op.progressive_slice(opt, cfg_p, sd_p, cfg_i, sd_i, torch.from_numpy(j["sino"])[None, None].to(dt),
                                   lambda: next(draws), sharpen_num=int(j["sharpen_num"]))
    assert next(draws, None) is None, "the oracle consumed fewer draws than the device recorded"
    if "mid" in j.files and int(j["mid"]):
        np.savez(out + ".mid.npz", proj=np.stack([m.numpy()[0, 0] for m in mid["proj"]]), fbp=mid["fbp"].numpy()[0, 0],
                 img=np.stack([m.numpy()[0, 0] for m in mid["img"]]))
    np.save(out, want.numpy())


if __name__ == "__main__":
    main()
