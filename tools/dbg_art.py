import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import art
from oracle import art as oa
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_art import _small, _phantoms
g, go, lut, betas = _small()
vol = _phantoms(3, g.nx)
proj = oa.project(go, lut, betas, vol)
plan = art.ArtPlan(lut, betas, device="cuda:0", geom=g)
gp = plan.project_device(torch.from_numpy(vol)).cpu().numpy()
print("project err", np.abs(gp - proj).max(), np.abs(proj).max())
for nsart, ntv in ((1, 0), (2, 0), (2, 1), (3, 2), (3, 0)):
    got = plan.reconstruct_device(torch.from_numpy(proj), nsart, ntv).cpu().numpy()
    want = oa.reconstruct(go, lut, betas, proj, nsart, ntv, permute=False)
    d = np.abs(got - want)
    i = np.unravel_index(np.argmax(d), d.shape)
    print(nsart, ntv, "err", d.max(), "at", i, got[i], want[i], "mean err", d.mean(), "count>1e-5", (d > 1e-5).sum())
