"""Option conv_bf16x3: the first forward of a process against the later ones (same input, same weights), both production UNets.
usage: dbg_bf16x3_fwd.py <option 0|1> [B]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ipdm_pytorch_amd import _lib, synth
from ipdm_pytorch_amd.config import default_cfg, cfg_load, mayo_test_options
from ipdm_pytorch_amd.denoiser import progressive_domain_denoiser
DEV = "cuda:0"
on = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
opt = default_cfg([])
cfg_load(mayo_test_options(), opt.__dict__)
cfg_load(dict(t_start_proj=[15, 15, 15], t_start_img=[15], ultra_img_denoise=True, device=DEV), opt.__dict__)
_lib.set_option("conv_bf16x3", on)
warm = os.environ.get("DBG_WARM", "")
if warm == "default":        # a forward of the float32 kernels first: every code object but conv_wino3's is loaded, scratch is set up
    _lib.set_option("conv_bf16x3", 0)
    d0 = progressive_domain_denoiser(opt, seed=1234)
    d0.proj_model.use_graph = False
    d0.proj_model(torch.from_numpy(synth.hash_normal((B, 1, 2000, 912), 5)).to(DEV), 7)
    torch.cuda.synchronize()
    del d0
    _lib.set_option("conv_bf16x3", on)
elif warm == "alloc":        # memory the process has touched before: 12 GiB filled and handed back to the caching allocator (the workspace, the
    blocks = [torch.full((256 << 20,), 1.0, device=DEV) for _ in range(12)]      # input and the outputs of the first forward come from these blocks)
    torch.cuda.synchronize()
    del blocks
elif warm == "scratch":      # the kernels of this library with the LARGEST private-segment (scratch) need first -- conv_ws: 256 ... 280 bytes per lane --, the host
    from oracle import unet as ou                                     # waiting behind each: the queue's scratch has its final size before the forward starts
    for (C, Co, H, W, ks, st) in [(16, 128, 64, 64, 3, 1), (128, 128, 16, 16, 3, 1), (128, 128, 64, 64, 3, 2), (128, 256, 64, 64, 1, 1), (16, 128, 250, 114, 3, 1)]:
        xw = torch.from_numpy(synth.hash_normal((1, C, H, W), 3)).to(DEV)
        Ho, Wo = (H + 2 * (ks // 2) - ks) // st + 1, (W + 2 * (ks // 2) - ks) // st + 1
        ow = torch.empty((1, Co, Ho, Wo), device=DEV)
        wn, bn, gn_, ben = (np.ascontiguousarray(t, dtype=np.float32) for t in (
            synth.hash_normal((Co, C, ks, ks), 5) / np.sqrt(C * ks * ks), synth.hash_normal((Co,), 6), synth.hash_uniform((C,), 7) + 0.5, synth.hash_normal((C,), 8) * 0.2))
        with _lib.option("conv_no_pw", 1):
            print("scratch warm-up: code", _lib.lib().ipdm_conv_kernel_code(1, Co, C, ks, st, H, W), (C, Co, H, W, ks, st), flush=True)
            _lib.call("ipdm_op_conv2d", _lib.ptr(xw), C, None, 0, 1, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), Co, ks, st,
                      0, ou.gn_groups(C), _lib.ptr(gn_), _lib.ptr(ben), None, _lib.ptr(ow), _lib.current_stream())
        torch.cuda.synchronize()
elif warm == "ops":          # conv_wino3 alone first (both non-planar instantiations), the host waiting behind each launch
    from oracle import unet as ou
    for res in (True, False):
        Bw, C, H, W = 2, 128, 64, 256
        xw = torch.from_numpy(synth.hash_normal((Bw, C, H, W), 3)).to(DEV)
        rw = torch.from_numpy(synth.hash_normal((Bw, C, H, W), 4)).to(DEV) if res else None
        ow = torch.empty((Bw, C, H, W), device=DEV)
        wn, bn, gn_, ben = (np.ascontiguousarray(t, dtype=np.float32) for t in (
            synth.hash_normal((C, C, 3, 3), 5) / np.sqrt(C * 9), synth.hash_normal((C,), 6), synth.hash_uniform((C,), 7) + 0.5, synth.hash_normal((C,), 8) * 0.2))
        assert _lib.lib().ipdm_conv_kernel_code(Bw, C, C, 3, 1, H, W) == (12 if on else 2)
        _lib.call("ipdm_op_conv2d", _lib.ptr(xw), C, None, 0, Bw, H, W, H, W, _lib.ptr(wn), _lib.ptr(bn), C, 3, 1,
                  2, ou.gn_groups(C), _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rw), _lib.ptr(ow), _lib.current_stream())
        torch.cuda.synchronize()
den = progressive_domain_denoiser(opt, seed=1234)
NETS = (("proj", den.proj_model, (2000, 912)), ("img", den.img_model, (512, 512)))
if os.environ.get("DBG_NETS"):
    NETS = [n for n in NETS if n[0] in os.environ["DBG_NETS"].split(",")]
for name, net, (H, W) in NETS:
    net.use_graph = False
    x = torch.from_numpy(synth.hash_normal((B, 1, H, W), 5)).to(DEV)
    outs = []
    for _ in range(4):
        outs.append(net(x, 7).cpu())
        if hasattr(_lib.lib(), "ipdm_trace_dump"):        # (libipdm_hip_trace2.so: the per-convolution device checksums of this forward)
            _lib.lib().ipdm_trace_dump()
    print("warm=%-7s option %d %s UNet B=%d: forwards 2..4 against the first: %s" % (warm, on, name, B, ["%.2e/%d" % ((o - outs[0]).abs().max().item(), int((o != outs[0]).sum())) for o in outs[1:]]), flush=True)
