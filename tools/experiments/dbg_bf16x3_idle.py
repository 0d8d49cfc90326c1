"""Option conv_bf16x3: is the wrong first forward a matter of the GPU waking up?  (a) a forward after the process has kept the GPU busy with
unrelated work (torch matmuls) -- nothing of this library has run before it; (b) forwards after idle pauses in a process whose forwards have
long been repeatable.  usage: dbg_bf16x3_idle.py <option 0|1> <prewarm seconds> [pauses...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ipdm_pytorch_amd import _lib, synth
from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser
DEV = "cuda:0"
on = int(sys.argv[1])
prewarm = float(sys.argv[2])
pauses = [float(v) for v in sys.argv[3:]]
opt = default_cfg([])
cfg_load(mayo_test_options(), opt.__dict__)
cfg_load(dict(t_start_proj=[15, 15, 15], t_start_img=[15], ultra_img_denoise=True, device=DEV), opt.__dict__)
_lib.set_option("conv_bf16x3", on)
x = torch.from_numpy(synth.hash_normal((2, 1, 2000, 912), 5)).to(DEV)
if prewarm > 0:
    a = torch.randn(8192, 8192, device=DEV)
    t0 = time.time()
    while time.time() - t0 < prewarm:
        for _ in range(10):
            a @ a
        torch.cuda.synchronize()
den = progressive_domain_denoiser(opt, seed=1234)
net = den.proj_model
net.use_graph = False
outs = [net(x, 7).cpu() for _ in range(4)]
ref = outs[-1]
print("option %d, %.0f s of matmuls first: forwards 1..3 against the fourth: %s" % (on, prewarm, ["%.2e" % (o - ref).abs().max().item() for o in outs[:3]]), flush=True)
for p in pauses:
    res = []
    for _ in range(4):
        time.sleep(p)
        res.append("%.2e" % (net(x, 7).cpu() - ref).abs().max().item())
    print("option %d: a forward after %.1f s of idle, four times: %s" % (on, p, res), flush=True)
