"""FBP: host-side mirror of Recon/FBP_kernel.py's FBP class; the arithmetic is libipdm_hip.so
(ipdm_fbp_forward: LDS-resident ramp filter + fp64-geometry pixel-driven back-projection)."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import FbpGeom, call, lib, ptr


# BASELINE.json's config C3 names 736 x 1152 sinograms: 1152 views of 736 detector channels (the reference's own geometry is
# fixed at 2000 x 912, Recon/FBP_kernel.py:34-40, so this shape has no reference counterpart -- perf / robustness only).
# Same source distance, FOV and total fan angle as the reference geometry (912 * 0.0010125 rad spread over 736 channels),
# 360 degrees in 1152 steps.
ALT_GEOMETRY = dict(n_views=1152, n_det=736, da=0.0012546, det_offset=3.0, dtheta_deg=0.3125, source_origin=59.5,
                    fov_half=21.0, grid_n=512)


class FBP:
    """FBP(device).convert(pj, flip=True) as in Recon/FBP_kernel.py:27-122.

    Geometry defaults are the reference's hard-coded values; they are constructor keywords here so
    that other sinogram shapes (BASELINE.json's perf-only 1152x736) can be planned too."""

    def __init__(self, device="cuda:0", n_views=2000, n_det=912, da=0.0010125, det_offset=3.75, dtheta_deg=0.18,
                 source_origin=59.5, fov_half=21.0, grid_n=512):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.IpdmError("FBP runs on the GPU only (no CPU fallback); got device=%r" % (device,))
        self.M, self.N, self.grid_n = n_views, n_det, grid_n     # reference names: M views, N detectors
        g = FbpGeom(n_views, n_det, grid_n, da, det_offset, dtheta_deg, source_origin, fov_half)
        self._geom = g
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            call("ipdm_fbp_plan_create", C.byref(g), C.byref(h))
        self._plan = h
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_plan", None) is not None:
                lib().ipdm_fbp_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    def table(self, which):
        """Host copy of a geometry table (0 theta,1 phi,2 r: f64; 3 nda,4 h_RL,5 weight: f32)."""
        n = lib().ipdm_fbp_table(self._plan, which, None, 0)
        out = np.empty(n, dtype=np.float64 if which < 3 else np.float32)
        lib().ipdm_fbp_table(self._plan, which, out.ctypes.data, n)
        return out

    def convert_device(self, pj, flip=True, gain=1.0):
        """[B, n_views, n_det] f32 cuda tensor -> [B, grid_n, grid_n] f32 cuda tensor (stays on device)."""
        if pj.dim() == 2:
            pj = pj[None]
        pj = pj.to(self.device, torch.float32).contiguous()
        B = pj.shape[0]
        if tuple(pj.shape[1:]) != (self.M, self.N):
            raise ValueError("sinogram shape %s does not match the plan (%d views x %d detectors)"
                             % (tuple(pj.shape), self.M, self.N))
        need = lib().ipdm_fbp_workspace_bytes(self._plan, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((B, self.grid_n, self.grid_n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            call("ipdm_fbp_forward", self._plan, ptr(pj), ptr(out), B, 1 if flip else 0, float(gain), ptr(self._ws),
                 self._ws.numel(), _lib.current_stream())
        return out

    def convert(self, pj, flip=True):
        """Reference semantics (Recon/FBP_kernel.py:86-122): Tensor in -> CPU Tensor out, ndarray in ->
        ndarray out."""
        is_tensor = isinstance(pj, torch.Tensor)
        t = pj if is_tensor else torch.from_numpy(np.ascontiguousarray(pj, dtype=np.float32))
        out = self.convert_device(t, flip=flip)
        return out.cpu() if is_tensor else out.cpu().numpy()

    def filter_device(self, pj, flip=True, gain=1.0):
        pj = pj.to(self.device, torch.float32).contiguous()
        out = torch.empty_like(pj)
        with torch.cuda.device(self.device):
            call("ipdm_fbp_filter", self._plan, ptr(pj), ptr(out), pj.shape[0], 1 if flip else 0, float(gain),
                 _lib.current_stream())
        return out

    def backproject_device(self, filtered, flip=False):
        f = filtered.to(self.device, torch.float32).contiguous()
        out = torch.empty((f.shape[0], self.grid_n, self.grid_n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            call("ipdm_fbp_backproject", self._plan, ptr(f), ptr(out), f.shape[0], 1 if flip else 0,
                 _lib.current_stream())
        return out

    def index_map(self, pixels):
        """u(t, p) float64 [n_views, len(pixels)] for flat pixel indices (parity of the index map)."""
        pix = torch.as_tensor(pixels, dtype=torch.int32, device=self.device).contiguous()
        out = torch.empty((self.M, pix.numel()), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            call("ipdm_fbp_index_map", self._plan, ptr(pix), pix.numel(), ptr(out), _lib.current_stream())
        return out


def tensor_sharpen(img_in, N=60):
    """Utils/train_test_utils.py:868-878 (per slice), on the GPU."""
    if N == -1:
        return img_in
    x = img_in.to(torch.float32).contiguous()
    if x.device.type != "cuda":
        raise _lib.IpdmError("tensor_sharpen runs on the GPU only")
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        call("ipdm_sharpen3x3", ptr(x), ptr(out), B * Cc, H, W, float(N), _lib.current_stream())
    return out
