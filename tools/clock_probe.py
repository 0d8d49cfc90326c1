"""The clock the chip holds under a kernel, with no profiler attached and no stamp in the kernel: a one-wave probe kernel
(ipdm_clock_probe, csrc/prof.hip) samples s_memtime / s_memrealtime on a stream of its own while the load runs.
   python tools/clock_probe.py fwd [proj|img] [B]                  whole UNet forwards back to back IN THIS PROCESS (the probe
                                                                    co-resides: a forward allocates nothing and never synchronises)
   python tools/clock_probe.py conv B C1 C2 H W Cout ks stride act res | attn B heads T | step
conv / attn / step run their load as a CHILD PROCESS (the micro-benchmark entries allocate and synchronise); two processes
TIME-SLICE the GPU -- the load runs at half speed and the probe's intervals blend loaded and idle slices: a lower bound of the
clock drop only.  `fwd` is the measurement."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ipdm_pytorch_amd import _lib

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
torch.zeros(1, device="cuda")
PERIOD_US = 50000 if mode != "step" else 250000
N = 300 if mode != "step" else 200
net = x = None
if mode == "fwd":
    from ipdm_pytorch_amd import synth
    from ipdm_pytorch_amd.unet import UNetModel
    from oracle import unet as ou
    which = sys.argv[2] if len(sys.argv) > 2 else "proj"
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    cfg, shape = ((ou.UNetConfig(), (B, 1, 512, 512)) if which == "img" else
                  (ou.UNetConfig(attention_resolutions=(16, 32), channel_mult=(1 / 16, 1 / 8, 1 / 4, 2, 2, 4, 4)), (B, 1, 2000, 912)))
    kw = {k: getattr(cfg, k) for k in ("in_channels", "model_channels", "out_channels", "num_res_blocks", "attention_resolutions",
                                       "channel_mult", "num_heads")}
    net = UNetModel(**kw).to("cuda")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(ou.param_shapes(cfg), seed=1).items()})
    x = torch.from_numpy(synth.hash_normal(shape, 3)).to("cuda")
    for _ in range(2):
        y = net(x, 7)                  # (workspace, packed weights: everything a forward needs exists after these)
    torch.cuda.synchronize()
    time.sleep(1.0)
out = torch.zeros(2 * N, dtype=torch.int64, device="cuda")
probe_stream = torch.cuda.Stream()
_lib.call("ipdm_clock_probe", out.data_ptr(), N, PERIOD_US, probe_stream.cuda_stream)
t0 = time.time()
time.sleep(0.6)                      # idle baseline first
CHILD = """
import ctypes as C, sys, os, time
sys.path.insert(0, %r)
import torch
from ipdm_pytorch_amd import _lib
torch.zeros(1, device="cuda")
ms = C.c_float()
args = [int(v) for v in sys.argv[2:]]
fn = "ipdm_bench_conv2d" if sys.argv[1] == "conv" else "ipdm_bench_attention"
if sys.argv[1] == "attn": args = [args[0], args[1], 64, args[2]]
_lib.call(fn, *args, 5, C.byref(ms))
iters = max(10, int(3000.0 / ms.value))
t = time.time()
_lib.call(fn, *args, iters, C.byref(ms))
print("LOAD %%.3f %%.3f %%s %%s: %%.3f ms per launch, %%d launches back to back" %% (t, time.time(), sys.argv[1], args, ms.value, iters))
""" % ROOT
if mode == "fwd":
    t_load0 = time.time() - t0
    n_fwd = 60 if which == "proj" else 100
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n_fwd):
        y = net(x, 7)
    ev1.record()
    ev1.synchronize()
    t_load1_abs = time.time() - t0
    what = "%s UNet forward, B = %d: %.2f ms per forward, %d forwards back to back in this process" % (which, B, ev0.elapsed_time(ev1) / n_fwd, n_fwd)
elif mode in ("conv", "attn"):
    r = subprocess.run([sys.executable, "-c", CHILD, mode] + sys.argv[2:], capture_output=True, text=True, cwd=ROOT)
    ln = [l for l in r.stdout.splitlines() if l.startswith("LOAD")]
    if not ln:
        print(r.stdout[-2000:], r.stderr[-2000:]); sys.exit(1)
    f = ln[0].split(" ", 3)
    t_load0, t_load1_abs, what = float(f[1]) - t0, float(f[2]) - t0, f[3]
else:
    t_load0 = time.time() - t0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-alt",
                        "--no-extra-legs", "--no-roofline"], capture_output=True, text=True, cwd=ROOT)
    what = "bench.py --steps 3 (child process): " + r.stdout.strip().splitlines()[-1][:160]
    t_load0 += 12.0          # (import, weights, warm-up step)
    t_load1_abs = time.time() - t0
t_load1 = t_load1_abs
probe_stream.synchronize()
v = out.cpu().numpy().astype("float64")
st, rt = v[0::2], v[1::2]
print(what)
print("load from %.2f s to %.2f s after the probe's launch; samples every %.0f ms:" % (t_load0, t_load1, PERIOD_US / 1e3))
clk = []
for i in range(1, N):
    dt = (rt[i] - rt[i - 1]) / 1e8
    ghz = (st[i] - st[i - 1]) / (rt[i] - rt[i - 1]) * 0.1
    tt = (rt[i] - rt[0]) / 1e8
    clk.append((tt, ghz))
print("GHz per interval (loaded ones): " + " ".join("%.2f" % g for t, g in clk if t_load0 - 0.2 < t < t_load1 + 0.2))
loaded = sorted(g for t, g in clk if t_load0 + 0.5 < t < t_load1 - 0.1)
idle = sorted(g for t, g in clk if t < t_load0 - 0.05)
if loaded:
    print("in-kernel clock under load (median of %d intervals, first 0.5 s dropped): %.3f GHz   min %.3f max %.3f   | idle before: %.3f GHz" %
          (len(loaded), loaded[len(loaded) // 2], loaded[0], loaded[-1], idle[len(idle) // 2] if idle else float("nan")))
