"""One process of bench.py's host-filling CPU baseline: the CPU oracle (test infrastructure, timed here as the BASELINE, never
on the product path) pinned to its own core set.   python tools/cpu_baseline_child.py <threads> <cpu,cpu,...> <budget_s>
Prints one JSON line {"img": seconds per img-UNet forward @512x512, "proj": seconds per proj-UNet forward @2000x912 or null}."""
import json
import os
import sys
import time

threads, cpus, budget = int(sys.argv[1]), [int(c) for c in sys.argv[2].split(",")], float(sys.argv[3])
try:
    os.sched_setaffinity(0, cpus)
except (AttributeError, OSError):
    pass
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                   # noqa: E402
from oracle import unet as ou                  # noqa: E402
from ipdm_pytorch_amd import synth             # noqa: E402

torch.set_num_threads(threads)
t_start = time.perf_counter()
cfg_i = ou.UNetConfig()
sd_i = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(ou.param_shapes(cfg_i), seed=1).items()}
x_i = torch.from_numpy(synth.hash_normal((1, 1, 512, 512), 3))
ou.unet_forward(cfg_i, sd_i, x_i[:, :, :128, :128], 7)       # warm-up: primitive creation, allocator
t0 = time.perf_counter()
ou.unet_forward(cfg_i, sd_i, x_i, 7)
img = time.perf_counter() - t0
proj = None
if (time.perf_counter() - t_start) + 2.2 * img < budget:      # a proj forward costs about twice an img forward
    cfg_p = ou.UNetConfig(attention_resolutions=(16, 32), channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4))
    sd_p = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(ou.param_shapes(cfg_p), seed=1).items()}
    x_p = torch.from_numpy(synth.hash_normal((1, 1, 2000, 912), 3))
    t0 = time.perf_counter()
    ou.unet_forward(cfg_p, sd_p, x_p, 7)
    proj = time.perf_counter() - t0
print(json.dumps({"img": img, "proj": proj}))
