"""Times the ART convertor at the reference's full geometry on the GPU box:  python tools/bench_art.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import art, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t0 = time.perf_counter()
plan = art.ArtPlan(art.area_lut(), art.view_angles(), device="cuda:0")
torch.cuda.synchronize(); print("plan create %.3f s" % (time.perf_counter() - t0))
mu = torch.from_numpy(np.stack([synth.rasterize(synth.ellipse_phantom(s)).astype(np.float32) for s in range(B)])).cuda()
for name, fn in (("project", lambda: plan.project_device(mu)),):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print("%s B=%d: %.4f s" % (name, B, time.perf_counter() - t0))
sino = out
for nstart, ntv in ((1, 0), (10, 0), (10, 3)):
    plan.reconstruct_device(sino, 1, 0); torch.cuda.synchronize()
    t0 = time.perf_counter(); rec = plan.reconstruct_device(sino, nstart, ntv); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rms = float(((rec - mu) ** 2).mean().sqrt() / mu.max())
    print("reconstruct B=%d nstart=%d ntv=%d: %.3f s  (%.1f us per view-launch, rel rms err %.4f)" % (B, nstart, ntv, dt, dt / (nstart * 2001) * 1e6, rms))
