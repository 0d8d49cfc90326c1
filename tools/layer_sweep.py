"""Per-layer cost model of one UNet forward on the GPU box: enumerates every conv / attention launch of the two
reference configurations (B slices), times each DISTINCT shape with the library's micro-benchmark entry points and
prints time x count, grouped by kernel family.   python tools/layer_sweep.py [B]"""
import ctypes as C, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd import _lib
from oracle.unet import UNetConfig, topology

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def launches(cfg, H, W):
    """(kind, C1, C2, H, W, Cout, ks, stride, act, res) per conv launch; ('attn', heads, T) per attention."""
    down, middle, up, ch_out = topology(cfg)
    out = []
    sizes = []

    def res(c1, c2, co, h, w):
        out.append(("conv", c1, c2, h, w, co, 3, 1, 2, 0))
        if c1 + c2 != co:
            out.append(("conv", c1, c2, h, w, co, 1, 1, 0, 0))
        out.append(("conv", co, 0, h, w, co, 3, 1, 2, 1))

    def attn(c, h, w):
        out.append(("conv", c, 0, h, w, 3 * c, 1, 1, 1, 0))
        out.append(("attn", cfg.num_heads, h * w))
        out.append(("conv", c, 0, h, w, c, 1, 1, 0, 1))

    h, w = H, W
    chans = []
    for layers in down:
        for l in layers:
            if l[0] == "conv":
                out.append(("conv", l[1], 0, h, w, l[2], 3, 1, 0, 0)); c = l[2]
            elif l[0] == "res":
                res(l[1], 0, l[2], h, w); c = l[2]
            elif l[0] == "attn":
                attn(l[1], h, w)
            elif l[0] == "down":
                out.append(("conv", l[1], 0, h, w, l[1], 3, 2, 0, 0)); h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        chans.append((c, h, w))
    for l in middle:
        if l[0] == "res": res(l[1], 0, l[2], h, w)
        else: attn(l[1], h, w)
    skip = chans.pop()
    for layers in up:
        this = skip
        if chans: skip = chans.pop()
        for li, l in enumerate(layers):
            if l[0] == "res":
                c2 = this[0] if li == 0 else 0
                res(l[1] - c2, c2, l[2], h, w); c = l[2]
            elif l[0] == "attn":
                attn(l[1], h, w)
            elif l[0] == "up":
                h, w = skip[1], skip[2]
                out.append(("conv", l[1], 0, h, w, l[1], 3, 1, 0, 0))
    out.append(("conv", ch_out, 0, h, w, cfg.out_channels, 3, 1, 2, 0))
    return out


torch.zeros(1, device="cuda")
ms = C.c_float()
cache = {}
fam = collections.OrderedDict()
for name, cfg, (H, W) in (("img", UNetConfig(), (512, 512)),
                          ("proj", UNetConfig(attention_resolutions=(16, 32), channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4)), (2000, 912))):
    cnt = collections.Counter(launches(cfg, H, W))
    rows = []
    for k, n in cnt.items():
        if k not in cache:
            if k[0] == "conv":
                _, c1, c2, h, w, co, ks, st, act, res = k
                if c2 and c1 % 8:       # the executor materialises such concats
                    c1, c2 = c1 + c2, 0
                best = 1e30
                for _ in range(2):          # (best of two calls: the first call of a shape pays allocation and clock ramp)
                    _lib.call("ipdm_bench_conv2d", B, c1, c2, h, w, co, ks, st, act, res, 5, C.byref(ms))
                    best = min(best, ms.value)
                ms.value = best
                ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
                fl = 2.0 * B * ho * wo * co * (k[1] + k[2]) * ks * ks
            else:
                _, heads, T = k
                best = 1e30
                for _ in range(2):
                    _lib.call("ipdm_bench_attention", B, heads, 64, T, 5, C.byref(ms))
                    best = min(best, ms.value)
                ms.value = best
                fl = 4.0 * B * heads * T * T * 64
            cache[k] = (ms.value, fl)
        t, fl = cache[k]
        rows.append((t * n, n, t, fl, k))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print("== %s UNet forward, B=%d: %.1f ms in conv/attention launches, %.1f TFLOP/s overall" % (name, B, total, sum(r[3] * r[1] for r in rows) / total / 1e9))
    for tt, n, t, fl, k in (rows if os.environ.get("SWEEP_ALL") else rows[:28]):
        print("  %6.2f ms (%4.1f%%)  %2d x %7.3f ms  %6.1f TF/s  %s" % (tt, 100 * tt / total, n, t, fl / t / 1e9, k))
    for tt, n, t, fl, k in rows:
        if k[0] == "attn": f = "attention"
        elif k[6] == 1: f = "conv1x1"
        elif k[7] == 2: f = "conv3x3 s2"
        elif k[5] <= 32: f = "conv3x3 narrow (Cout<=32)"
        else: f = "conv3x3 wide"
        a = fam.setdefault((name, f), [0.0, 0.0])
        a[0] += tt; a[1] += fl * n
for (name, f), (tt, fl) in fam.items():
    print("family %-5s %-28s %8.2f ms  %6.1f TF/s" % (name, f, tt, fl / tt / 1e9))
