"""Child of tests/test_gpu_parity.py::test_bf16x3_first_forward_of_a_process: in a FRESH process, the production projection UNet under option
conv_bf16x3, one input, four forwards -- the first one (code objects loaded on the way, launches arriving behind the host's first-use work)
against the later ones, bit for bit.  Exit code 3: they differ."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ipdm_pytorch_amd import _lib, synth  # noqa: E402
from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options  # noqa: E402
from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser  # noqa: E402

opt = default_cfg([])
cfg_load(mayo_test_options(), opt.__dict__)
cfg_load(dict(t_start_proj=[15, 15, 15], t_start_img=[15], ultra_img_denoise=True, device="cuda:0"), opt.__dict__)
_lib.set_option("conv_bf16x3", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
den = progressive_domain_denoiser(opt, seed=1234)
net = den.proj_model
net.use_graph = False
x = torch.from_numpy(synth.hash_normal((2, 1, 2000, 912), 5)).to("cuda:0")
outs = [net(x, 7).cpu() for _ in range(4)]
d = [(o - outs[-1]).abs().max().item() for o in outs[:3]]
print("first forwards against the fourth: %s" % ["%.2e" % v for v in d])
sys.exit(3 if any(d) else 0)
