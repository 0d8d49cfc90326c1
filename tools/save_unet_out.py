"""Dumps one forward of the two production UNets (true sizes, B = 1, seeded weights and input) in the mode the environment
selects, for the fp64 accuracy study of tools/unet_fp64_study.py:  python tools/save_unet_out.py <tag>  -> gpurun_out/unet_<which>_<tag>.npy"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import synth
from ipdm_pytorch_amd.unet import UNetModel
tag = sys.argv[1]
FULL_IMG = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(8, 16), channel_mult=(1, 1, 2, 2, 4, 4))
FULL_PROJ = dict(in_channels=1, model_channels=64, out_channels=1, attention_resolutions=(16, 32),
                 channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4))
os.makedirs("gpurun_out", exist_ok=True)
for which, kw, shape in (("img", FULL_IMG, (1, 1, 512, 512)), ("proj", FULL_PROJ, (1, 1, 2000, 912))):
    net = UNetModel(**kw).to("cuda:0")
    sd = synth.synth_state_dict(net._shapes, seed=6)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = torch.from_numpy(synth.hash_normal(shape, 401))
    out = net(x.to("cuda:0"), 13).cpu().numpy()
    np.save("gpurun_out/unet_%s_%s.npy" % (which, tag), out)
    print(which, tag, out.shape, float(np.abs(out).max()))
