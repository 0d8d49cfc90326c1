#!/usr/bin/env python
"""Per-op accuracy against float64: for each kernel family, rms / max error of the HIP result and of torch-CPU float32 on
the same float32 inputs, both measured against a float64 evaluation.  (The end-to-end arbiter of tests/test_gpu_parity.py
says whether the library as a whole is as accurate as the CPU path; this says which op to look at when it is not.)

  python tools/accuracy_vs_f64.py            (GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                    # noqa: E402
import torch                          # noqa: E402
import torch.nn.functional as F       # noqa: E402
from ipdm_pytorch_amd import _lib, synth   # noqa: E402
from oracle import unet as ou         # noqa: E402

DEV = "cuda:0"


def dist(a, ref):
    e = (a.double() - ref).abs()
    return float(e.max()), float((e ** 2).mean().sqrt())


def report(name, hip, cpu32, ref):
    hm, hr = dist(hip, ref)
    cm, cr = dist(cpu32, ref)
    print("%-58s hip max %.2e rms %.2e | torch32 max %.2e rms %.2e | ratio max %.2f rms %.2f" % (
        name, hm, hr, cm, cr, hm / max(cm, 1e-30), hr / max(cr, 1e-30)), flush=True)


def conv_case(B, C1, Hs, Ws, Cout, ks, stride, act, res, up=False, seed=1):
    H, W = (2 * Hs, 2 * Ws) if up else (Hs, Ws)
    x = torch.from_numpy(synth.hash_normal((B, C1, Hs, Ws), seed))
    w = torch.from_numpy(synth.hash_normal((Cout, C1, ks, ks), seed + 2)) / np.sqrt(C1 * ks * ks)
    bias = torch.from_numpy(synth.hash_normal((Cout,), seed + 3))
    gamma = torch.from_numpy(synth.hash_uniform((C1,), seed + 4)) + 0.5
    beta = torch.from_numpy(synth.hash_normal((C1,), seed + 5)) * 0.2
    groups = ou.gn_groups(C1)

    def ref(dt):
        h = x.to(dt)
        if act:
            h = F.group_norm(h, groups, gamma.to(dt), beta.to(dt), eps=1e-5)
            if act == 2:
                h = F.silu(h)
        if up:
            h = F.interpolate(h, size=(H, W), mode="nearest")
        o = F.conv2d(h, w.to(dt), bias.to(dt), stride=stride, padding=ks // 2)
        return o
    want64, want32 = ref(torch.float64), ref(torch.float32)
    r = torch.from_numpy(synth.hash_normal(tuple(want32.shape), seed + 6)) if res else None
    if res:
        want64, want32 = want64 + r.double(), want32 + r
    out = torch.empty(tuple(want32.shape), device=DEV)
    x1d = x.to(DEV)
    rd = r.to(DEV) if res else None
    wn, bn, gn_, ben = (np.ascontiguousarray(t.numpy()) for t in (w, bias, gamma, beta))
    _lib.call("ipdm_op_conv2d", _lib.ptr(x1d), C1, None, 0, B, Hs, Ws, H, W, _lib.ptr(wn), _lib.ptr(bn), Cout, ks,
              stride, act, groups, _lib.ptr(gn_), _lib.ptr(ben), _lib.ptr(rd), _lib.ptr(out), _lib.current_stream())
    report("conv %dx%d s%d %d->%d @%dx%d act%d res%d up%d" % (ks, ks, stride, C1, Cout, Hs, Ws, act, int(res), int(up)),
           out.cpu(), want32, want64)


def attn_case(B, heads, T):
    d = 64
    qkv = torch.from_numpy(synth.hash_normal((B, heads * 3 * d, T), 300 + T)) * 1.5
    out = torch.empty((B, heads * d, T), device=DEV)
    _lib.call("ipdm_op_attention", _lib.ptr(qkv.to(DEV)), _lib.ptr(out), B, heads, d, T, _lib.current_stream())

    def ref(dt):
        q, k, v = qkv.to(dt).reshape(B * heads, 3 * d, T).chunk(3, dim=1)
        scale = 1.0 / np.sqrt(np.sqrt(d))
        a = torch.einsum("bct,bcs->bts", q * scale, k * scale).softmax(dim=-1)
        return torch.einsum("bts,bcs->bct", a, v).reshape(B, heads * d, T)
    report("attention B%d heads%d T%d" % (B, heads, T), out.cpu(), ref(torch.float32), ref(torch.float64))


def unet_case(tag, kw, shape, t):
    from ipdm_pytorch_amd.unet import UNetModel
    net = UNetModel(**kw).to(DEV)
    sd = synth.synth_state_dict(net._shapes, seed=21)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = torch.from_numpy(synth.hash_normal(shape, 401))
    got = net(x.to(DEV), t).cpu()
    cfg = ou.UNetConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()})
    w32 = ou.unet_forward(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, x, t)
    w64 = ou.unet_forward(cfg, {k: torch.from_numpy(v).double() for k, v in sd.items()}, x.double(), t)
    report("unet %s %s t=%d" % (tag, shape, t), got, w32, w64)
    for name in ("conv_no_up2", "gn_unfused"):
        with _lib.option(name, 1):
            report("   same with %s=1" % name, net(x.to(DEV), t).cpu(), w32, w64)


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    conv_case(1, 64, 64, 64, 64, 3, 1, 0, False)
    conv_case(1, 64, 64, 64, 64, 3, 1, 2, False)
    conv_case(1, 64, 64, 64, 64, 3, 1, 2, True)
    conv_case(1, 128, 64, 64, 128, 3, 1, 2, True)
    conv_case(1, 256, 32, 32, 256, 3, 1, 2, True)
    conv_case(1, 128, 32, 32, 128, 3, 1, 0, False, up=True)
    conv_case(1, 256, 32, 32, 768, 1, 1, 1, False)
    conv_case(1, 64, 64, 64, 64, 3, 2, 0, False)
    conv_case(1, 8, 128, 128, 8, 3, 1, 2, True)
    conv_case(1, 16, 128, 128, 16, 3, 1, 2, False)
    conv_case(1, 144, 64, 64, 16, 3, 1, 2, False)
    attn_case(1, 4, 1024)
    attn_case(1, 4, 4096)
    attn_case(1, 1, 7125)
    from ipdm_pytorch_amd.denoiser import SMOKE_PROJ, SMOKE_IMG
    unet_case("smoke-img", SMOKE_IMG, (1, 1, 512, 512), 1)
    unet_case("smoke-proj", SMOKE_PROJ, (1, 1, 2000, 912), 1)


if __name__ == "__main__":
    main()
