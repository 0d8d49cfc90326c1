"""CPU oracle: the "ART" domain convertor (SART over a triangle-area lookup table) and its forward projector --
Recon/TASART2DNSL0-Cpp/TASART2DNSL0.cu driven by TASART2DNSL0_PyAPI.cpp (recons_torch / proj_torch).

TEST INFRASTRUCTURE ONLY -- imported by tests/ (and dev-time checks); never by the product path.
PARITY UNPINNED for the reconstruction / projection arithmetic (the CUDA reference cannot run here, see
art_oracle.c); the area table and the view angles ARE pinned on the reference's data files
(`python oracle/art.py` in the build container: area_lut() vs Recon/Simens_alut.txt, view_angles() vs
Recon/Simens_theta.txt).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Geom(ctypes.Structure):
    """Parameters, TASART2DNSL0.h:23-42; defaults = TASART2DNSL0_PyAPI.cpp:9-28."""
    _fields_ = [("dso", ctypes.c_float), ("dsd", ctypes.c_float), ("nx", ctypes.c_int), ("ny", ctypes.c_int),
                ("dx", ctypes.c_float), ("dy", ctypes.c_float), ("offset_x", ctypes.c_float), ("offset_y", ctypes.c_float),
                ("nr", ctypes.c_int), ("dr", ctypes.c_float), ("offset_r", ctypes.c_float), ("angle_start", ctypes.c_float),
                ("na", ctypes.c_int), ("ta_dimx", ctypes.c_int), ("ta_dimy", ctypes.c_int),
                ("ta_deltax", ctypes.c_float), ("ta_deltay", ctypes.c_float)]


def geometry(nx=512, nr=912, na=2000, fov=42.0, dr=0.0010125, offset_r=-3.75, dso=59.5, dsd=108.56):
    f = np.float32
    dx = f(fov) / f(nx)
    ta_dx = dx * np.sqrt(f(2.0)) * f(0.5) / f(1500.0)
    return Geom(dso, dsd, nx, nx, dx, dx, 0.0, 0.0, nr, dr, offset_r, 0.0, na, 1501, 181, ta_dx, f(45.0) / f(180.0))


def area_lut(pixel, dimx=1501, dimy=181):
    """The table Recon/Simens_alut.txt holds: area of a pixel square (side `pixel`) beyond a line at distance
    d = i * (half diagonal / (dimx-1)) from its centre, ray direction j * 45/(dimy-1) degrees; [dimy, dimx] f32."""
    a = float(pixel)
    h = a / 2
    d = np.arange(dimx)[None, :] * (a * np.sqrt(2.0) * 0.5 / (dimx - 1))
    th = np.deg2rad(np.arange(dimy) * 45.0 / (dimy - 1))[:, None]
    c, s = np.cos(th), np.sin(th)
    t1, t2 = h * (c - s), h * (c + s)
    with np.errstate(divide="ignore", invalid="ignore"):
        tri = (t2 - d) ** 2 / (2 * c * s)
        full = (t1 - d) * (a / c) + (t2 - t1) ** 2 / (2 * c * s)
    out = np.where(d >= t2, 0.0, np.where(d >= t1, tri, full))
    out[0, :] = np.maximum(h - d[0], 0) * a
    return np.ascontiguousarray(out.astype(np.float32))


def view_angles(na=2000, step=0.18):
    """Recon/Simens_theta.txt: view angles in degrees, f32."""
    return (np.arange(na) * np.float64(step)).astype(np.float32)


def _lib():
    path = os.path.join(_HERE, "libipdm_oracle.so")
    if not os.path.isfile(path):
        raise RuntimeError("oracle/libipdm_oracle.so missing: run `make -C oracle`")
    return ctypes.CDLL(path)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def project(geom, lut, betas, vol):
    """proj_torch: vol [B, ny, nx] -> [B, na, nr]."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    betas = np.ascontiguousarray(betas, dtype=np.float32)
    out = np.zeros((vol.shape[0], geom.na, geom.nr), dtype=np.float32)
    for b in range(vol.shape[0]):
        _lib().art_oracle_project(ctypes.byref(geom), _fp(lut), _fp(betas), _fp(vol[b]), _fp(out[b]))
    return out


def reconstruct(geom, lut, betas, proj, nsart, ntv, permute=True):
    """recons_torch (sample_rate=1): proj [B, na, nr] -> [B, ny, nx] (transposed when permute, PyAPI.cpp:55-57)."""
    proj = np.ascontiguousarray(proj, dtype=np.float32)
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    betas = np.ascontiguousarray(betas, dtype=np.float32)
    out = np.zeros((proj.shape[0], geom.ny, geom.nx), dtype=np.float32)
    for b in range(proj.shape[0]):
        _lib().art_oracle_reconstruct(ctypes.byref(geom), _fp(lut), _fp(betas), _fp(proj[b]), _fp(out[b]),
                                      int(nsart), int(ntv))
    return out.transpose(0, 2, 1) if permute else out


if __name__ == "__main__":      # dev-time pin of the two data tables against the reference's files
    ref = "/root/reference/Recon/"
    sa = np.fromfile(ref + "Simens_alut.txt", "float32").reshape(181, 1501)
    st = np.fromfile(ref + "Simens_theta.txt", "float32")
    lut = area_lut(np.float32(42.0) / np.float32(512.0))
    print("area table: max |generated - Simens_alut| = %.3e (max value %.3e)" % (np.abs(lut - sa).max(), sa.max()))
    print("view angles: max |generated - Simens_theta| = %.3e" % np.abs(view_angles() - st).max())
