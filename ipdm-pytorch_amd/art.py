"""The "ART" domain convertor and the forward projector: host-side mirror of Recon/TASART2DNSL0's
recons_torch / proj_torch (TASART2DNSL0.pyi, TASART2DNSL0_PyAPI.cpp:33-89); the arithmetic is
libipdm_hip.so (ipdm_art_reconstruct / ipdm_art_project, csrc/art.hip).  SURVEY section 8(f) rank 3.

The two data files the reference reads at start-up (Utils/train_test_utils.py:226-227) can be passed in as
arrays exactly as the reference does, or regenerated: `area_lut()` reproduces Recon/Simens_alut.txt (the area
of a pixel square beyond a line, by distance and ray direction) to 1e-17 and `view_angles()` reproduces
Recon/Simens_theta.txt exactly."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import call, lib, ptr

_f32 = C.c_float
_i32 = C.c_int32


class ArtGeom(C.Structure):
    """ipdm_art_geom = Parameters (TASART2DNSL0.h:23-42)."""
    _fields_ = [("dso", _f32), ("dsd", _f32), ("nx", _i32), ("ny", _i32), ("dx", _f32), ("dy", _f32),
                ("offset_x", _f32), ("offset_y", _f32), ("nr", _i32), ("dr", _f32), ("offset_r", _f32),
                ("angle_start", _f32), ("na", _i32), ("ta_dimx", _i32), ("ta_dimy", _i32), ("ta_deltax", _f32),
                ("ta_deltay", _f32)]


def default_geom(nx=512, nr=912, na=2000, fov=42.0, dr=0.0010125, offset_r=-3.75, dso=59.5, dsd=108.56):
    """The hard-coded `params` of TASART2DNSL0_PyAPI.cpp:9-28 (keywords for other grids)."""
    f = np.float32
    dx = f(fov) / f(nx)
    ta_dx = dx * np.sqrt(f(2.0)) * f(0.5) / f(1500.0)
    return ArtGeom(dso, dsd, nx, nx, dx, dx, 0.0, 0.0, nr, dr, offset_r, 0.0, na, 1501, 181, ta_dx, f(45.0) / f(180.0))


def area_lut(pixel=np.float32(42.0) / np.float32(512.0), dimx=1501, dimy=181):
    """Area of a square pixel (side `pixel`) beyond a line at distance i * half_diagonal / (dimx - 1) from its centre,
    for ray directions j * 45 / (dimy - 1) degrees: [dimy, dimx] float32."""
    a = float(pixel)
    h = a / 2
    d = np.arange(dimx)[None, :] * (a * np.sqrt(2.0) * 0.5 / (dimx - 1))
    th = np.deg2rad(np.arange(dimy) * 45.0 / (dimy - 1))[:, None]
    c, s = np.cos(th), np.sin(th)
    t1, t2 = h * (c - s), h * (c + s)
    with np.errstate(divide="ignore", invalid="ignore"):
        corner = (t2 - d) ** 2 / (2 * c * s)                      # the line cuts off one corner
        band = (t1 - d) * (a / c) + (t2 - t1) ** 2 / (2 * c * s)  # ... or crosses two opposite sides
    out = np.where(d >= t2, 0.0, np.where(d >= t1, corner, band))
    out[0, :] = np.maximum(h - d[0], 0) * a                        # axis-parallel rays
    return np.ascontiguousarray(out.astype(np.float32))


def view_angles(na=2000, step=0.18):
    return (np.arange(na) * np.float64(step)).astype(np.float32)


class ArtPlan:
    """Device-side plan: geometry, area table, rays of every view, normalisation projection."""

    def __init__(self, lut_area, betas, device="cuda:0", geom=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.IpdmError("the ART convertor runs on the GPU only (no CPU fallback); got device=%r" % (device,))
        betas = np.ascontiguousarray(np.asarray(betas, dtype=np.float32).reshape(-1))
        g = geom if geom is not None else default_geom(na=betas.size)
        if g.na != betas.size:
            raise ValueError("geometry has %d views, betas %d" % (g.na, betas.size))
        lut = np.ascontiguousarray(np.asarray(lut_area, dtype=np.float32).reshape(-1))
        if lut.size != g.ta_dimx * g.ta_dimy:
            raise ValueError("lut_area has %d entries, geometry wants %d x %d" % (lut.size, g.ta_dimy, g.ta_dimx))
        self.geom = g
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            call("ipdm_art_plan_create", C.byref(g), ptr(lut), ptr(betas), C.byref(h))
        self._plan = h
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_plan", None) is not None:
                lib().ipdm_art_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    def _workspace(self, B):
        need = lib().ipdm_art_workspace_bytes(self._plan, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def reconstruct_device(self, proj, nstart, ntv, sample_rate=1):
        """[B, na, nr] -> [B, ny, nx] f32 on the device (not permuted)."""
        g = self.geom
        p = proj.to(self.device, torch.float32).contiguous()
        if p.dim() != 3 or tuple(p.shape[1:]) != (g.na, g.nr):
            raise ValueError("projection shape %s does not match the plan (%d views x %d bins)" % (tuple(p.shape), g.na, g.nr))
        B = p.shape[0]
        out = torch.empty((B, g.ny, g.nx), dtype=torch.float32, device=self.device)
        ws = self._workspace(B)
        with torch.cuda.device(self.device):
            call("ipdm_art_reconstruct", self._plan, ptr(p), ptr(out), B, int(nstart), int(ntv), int(sample_rate), ptr(ws),
                 ws.numel(), _lib.current_stream())
        return out

    def project_device(self, volume):
        """[B, ny, nx] -> [B, na, nr] f32 on the device."""
        g = self.geom
        v = volume.to(self.device, torch.float32).contiguous()
        if v.dim() != 3 or tuple(v.shape[1:]) != (g.ny, g.nx):
            raise ValueError("volume shape %s does not match the plan (%d x %d)" % (tuple(v.shape), g.ny, g.nx))
        B = v.shape[0]
        out = torch.empty((B, g.na, g.nr), dtype=torch.float32, device=self.device)
        ws = self._workspace(B)
        with torch.cuda.device(self.device):
            call("ipdm_art_project", self._plan, ptr(v), ptr(out), B, ptr(ws), ws.numel(), _lib.current_stream())
        return out


_PLANS = {}


def _plan_for(lut_area, betas, device):
    """recons_torch / proj_torch take the tables on every call; the plan built from them is cached by content."""
    lut = np.ascontiguousarray(np.asarray(lut_area, dtype=np.float32))
    bet = np.ascontiguousarray(np.asarray(betas, dtype=np.float32))
    key = (str(torch.device(device)), lut.size, bet.size, hash(lut.tobytes()), hash(bet.tobytes()))
    if key not in _PLANS:
        _PLANS[key] = ArtPlan(lut, bet, device=device)
    return _PLANS[key]


def _device_of(t, device):
    if device is not None:
        return torch.device(device)
    return t.device if t.device.type == "cuda" else torch.device("cuda:0")


def recons_torch(h_proj, lut_area, betas, nstart, ntv, sample_rate=1, permute=True, device=None):
    """Same call as Recon/TASART2DNSL0.recons_torch (TASART2DNSL0.pyi:5-16): h_proj [B, 2000, 912] -> [B, 512, 512],
    transposed when `permute` (a view, PyAPI.cpp:55-57).  A CPU tensor comes back on the CPU (the reference returns a
    host tensor), a CUDA tensor stays on its device."""
    plan = _plan_for(lut_area, betas, _device_of(h_proj, device))
    out = plan.reconstruct_device(h_proj, nstart, ntv, sample_rate)
    if permute:
        out = out.permute(0, 2, 1)
    return out if h_proj.device.type == "cuda" else out.cpu()


def proj_torch(h_volume, lut_area, betas, device=None):
    """Same call as Recon/TASART2DNSL0.proj_torch (TASART2DNSL0.pyi:18-24): h_volume [B, 512, 512] -> [B, 2000, 912]."""
    plan = _plan_for(lut_area, betas, _device_of(h_volume, device))
    out = plan.project_device(h_volume)
    return out if h_volume.device.type == "cuda" else out.cpu()
