"""UNetModel: host-side mirror of the reference's Model/model.py:190-310 (same constructor
arguments, same state_dict key layout) whose forward runs entirely in libipdm_hip.so
(ipdm_unet_forward: implicit-GEMM convs on the f32 MFMA, GroupNorm statistics, flash attention).

torch is used only to hold the parameters (so `state_dict()` / `load_state_dict()` of reference
checkpoints work, Utils/loggerx.py:62-80) and to own the device workspace.
"""
import collections
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import UnetCfg, call, lib, ptr


def make_cfg(in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, channel_mult,
             num_heads):
    cfg = UnetCfg()
    cfg.in_channels, cfg.model_channels, cfg.out_channels = in_channels, model_channels, out_channels
    cfg.num_res_blocks, cfg.num_heads = num_res_blocks, num_heads
    cfg.n_mult, cfg.n_attn = len(channel_mult), len(attention_resolutions)
    if cfg.n_mult > 16 or cfg.n_attn > 16:
        raise ValueError("channel_mult / attention_resolutions longer than 16")
    for i, m in enumerate(channel_mult):
        cfg.channel_mult[i] = float(m)
    for i, a in enumerate(attention_resolutions):
        cfg.attention_resolutions[i] = int(a)
    return cfg


def param_shapes(cfg):
    """Ordered {state_dict key: shape} as the native library expects them (== the reference's
    UNetModel.state_dict() layout)."""
    n = lib().ipdm_unet_param_count(C.byref(cfg))
    if n < 0:
        raise _lib.IpdmError(lib().ipdm_last_error().decode())
    out = collections.OrderedDict()
    name = C.create_string_buffer(128)
    shape = (C.c_int32 * 4)()
    nd = C.c_int32()
    for i in range(n):
        call("ipdm_unet_param_info", C.byref(cfg), i, name, 128, C.byref(shape), C.byref(nd))
        out[name.value.decode()] = tuple(shape[k] for k in range(nd.value))
    return out


class UNetModel:
    """Drop-in for the reference's UNetModel on the sampling path (inference only)."""

    def __init__(self, in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2,
                 attention_resolutions=(8, 16), dropout=0, channel_mult=(1, 2, 2, 2), conv_resample=True,
                 num_heads=4, pre_downsample_times=1):
        if dropout:
            raise NotImplementedError("dropout is never instantiated on the sampling path (Model/model.py:198)")
        if not conv_resample:
            raise NotImplementedError("conv_resample=False (AvgPool down-sampling) is not on the reference's path")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        self.channel_mult = tuple(channel_mult)
        self.num_heads = num_heads
        self.cfg = make_cfg(in_channels, model_channels, out_channels, num_res_blocks, self.attention_resolutions,
                            self.channel_mult, num_heads)
        self._shapes = param_shapes(self.cfg)
        # default initialisation: deterministic synthetic weights (there is no training here)
        from . import synth
        self._params = collections.OrderedDict(
            (k, torch.from_numpy(v)) for k, v in synth.synth_state_dict(self._shapes, seed=0).items())
        self._device = torch.device("cpu")
        self._handle = None
        self._ws = None
        # graph replay (ipdm_unet_forward_graph): forwards go through static input / output buffers so that the captured
        # launches can be re-issued; IPDM_UNET_GRAPH=1 or `model.use_graph = True`
        self.use_graph = os.environ.get("IPDM_UNET_GRAPH", "0") not in ("", "0")
        self._gbuf = {}

    # ---- nn.Module-like surface used by the reference harness
    def state_dict(self):
        return collections.OrderedDict((k, v.clone()) for k, v in self._params.items())

    def load_state_dict(self, sd, strict=True):
        sd = {k.replace("module.", ""): v for k, v in sd.items()}   # load_network, Utils/loggerx.py:131-140
        missing = [k for k in self._shapes if k not in sd]
        extra = [k for k in sd if k not in self._shapes]
        if strict and (missing or extra):
            raise RuntimeError("load_state_dict: missing %s unexpected %s" % (missing[:5], extra[:5]))
        for k, shp in self._shapes.items():
            if k in sd:
                v = torch.as_tensor(sd[k]).detach().to(torch.float32).cpu().contiguous()
                if tuple(v.shape) != tuple(shp):
                    raise RuntimeError("load_state_dict: %s has shape %s, expected %s" % (k, tuple(v.shape), shp))
                self._params[k] = v
        self._destroy()
        return self

    def parameters(self):
        dev = self._device
        for v in self._params.values():
            yield v if dev.type == "cpu" else _DeviceView(v, dev)

    def to(self, device):
        self._device = torch.device(device)
        return self

    def eval(self):
        return self

    # ---- native handle
    def _destroy(self):
        if self._handle is not None:
            lib().ipdm_unet_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _ensure(self):
        if self._handle is None:
            if self._device.type != "cuda":
                raise _lib.IpdmError("UNetModel runs on the GPU only (no CPU fallback): call .to('cuda:N')")
            arr = (C.c_void_p * len(self._shapes))()
            keep = []
            for i, k in enumerate(self._shapes):
                a = np.ascontiguousarray(self._params[k].numpy(), dtype=np.float32)
                keep.append(a)
                arr[i] = a.ctypes.data
            h = C.c_void_p()
            with torch.cuda.device(self._device):
                call("ipdm_unet_create", C.byref(self.cfg), arr, len(keep), C.byref(h))
            self._handle = h
        return self._handle

    def workspace(self, B, H, W):
        need = lib().ipdm_unet_workspace_bytes(self._ensure(), B, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self._device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self._device)
        return self._ws

    def forward_into(self, x, t, out):
        """x [B,Cin,H,W] f32 cuda contiguous, integer timestep t -> out [B,Cout,H,W]."""
        B, _, H, W = x.shape
        ws = self.workspace(B, H, W)
        with torch.cuda.device(self._device):
            call("ipdm_unet_forward", self._ensure(), ptr(x), int(t), ptr(out), B, H, W, ptr(ws), ws.numel(),
                 _lib.current_stream())
        return out

    def __call__(self, x, timesteps):
        """UNetModel.forward(x, timesteps) (Model/model.py:283): one timestep for the whole batch."""
        t = timesteps
        if isinstance(t, torch.Tensor):
            vals = t.reshape(-1).tolist()
            if any(v != vals[0] for v in vals):
                raise NotImplementedError("per-sample timesteps: the sampling path always passes one t (model.py:564)")
            t = int(vals[0])
        x = x.to(self._device, torch.float32).contiguous()
        if self.use_graph:
            return self._forward_graph(x, t)
        out = torch.empty((x.shape[0], self.out_channels, x.shape[2], x.shape[3]), dtype=torch.float32,
                          device=self._device)
        return self.forward_into(x, t, out)

    def _forward_graph(self, x, t):
        """The forward replayed from a hipGraph: x is copied into a static buffer, the captured launches write a static
        output buffer, and a copy of it is returned (one extra pass over [B,1,H,W] each way beside ~400 launches)."""
        B, _, H, W = x.shape
        key = (B, H, W)
        buf = self._gbuf.get(key)
        if buf is None or buf[0].device != self._device:
            buf = (torch.empty_like(x), torch.empty((B, self.out_channels, H, W), dtype=torch.float32, device=self._device))
            self._gbuf = {key: buf}              # one shape at a time: the buffers (and the graphs keyed on them) are large
        gx, gout = buf
        ws = self.workspace(B, H, W)
        with torch.cuda.device(self._device):
            gx.copy_(x)
            call("ipdm_unet_forward_graph", self._ensure(), ptr(gx), int(t), ptr(gout), B, H, W, ptr(ws), ws.numel(),
                 _lib.current_stream())
            return gout.clone()

    forward = __call__


class _DeviceView:
    """What `next(model.parameters())` must answer on the reference's harness: .device and .dtype."""

    def __init__(self, t, device):
        self.device, self.dtype, self.shape = device, t.dtype, t.shape
