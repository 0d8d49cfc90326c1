"""CPU oracle: fan-beam FBP domain convertor (Recon/FBP_kernel.py:27-184), geometry parametrised.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product path.

Geometry/constants are restated in numpy exactly as FBP.__init__ / getrphi write them (same
expressions, same dtypes) so that for the reference geometry (2000 views x 912 detectors ->
512x512) theta, nda, h_RL, r, phi are identical to the imported reference (checked by
oracle/check_vs_reference.py and tests/golden/fbp_geometry.npz).  The two hot loops live in
fbp_oracle.c (libipdm_oracle.so, built by oracle/Makefile).

numpy-version note: FBP.convert multiplies the float32 sinogram by the float64 numpy scalar
(theta[1]-theta[0]).  Under the reference's pinned numpy 1.19 (requirements.txt) value-based
casting keeps that product in float32; numpy>=2 (this image) promotes to float64 before the final
astype(float32).  The oracle follows the reference's pinned environment (float32 product); the two
differ by at most 1 ulp of float32 per sample.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libipdm_oracle.so")
        if not os.path.isfile(path):
            raise RuntimeError("oracle/libipdm_oracle.so missing: run `make -C oracle` "
                               "(or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
    return _LIB


class FBPGeometry:
    """FBP.__init__ + getrphi, Recon/FBP_kernel.py:28-84.  Defaults = reference values."""

    def __init__(self, n_views=2000, n_det=912, da=0.0010125, det_offset=3.75, dtheta_deg=0.18,
                 source_origin=59.5, fov_half=21.0, grid_n=512):
        self.n_views, self.n_det, self.da = n_views, n_det, da
        self.D = abs(-source_origin)
        self.grid_n, self.fov_half = grid_n, fov_half
        # :38  np.arange(0, 359.82+0.18, 0.18)/180*pi  (arange value i == start + i*step)
        self.theta = (np.arange(n_views) * dtheta_deg) / 180 * np.pi
        # :39-40  np.arange(start, stop, da).astype(f32); numpy fills start + i*((start+da)-start)
        start = (-n_det / 2 + 0.5 + det_offset) * da
        delta = (start + da) - start
        self.nda = (start + np.arange(n_det) * delta).astype("float32")
        # :52-56 ramp kernel
        h = np.zeros((2 * n_det - 1, 1))
        ngarma = np.arange(-n_det + 1, n_det, 2) * da
        h[0:2 * n_det - 1:2] = (-0.5 / np.pi ** 2. / (np.sin(ngarma) ** 2))[:, None]
        h[n_det - 1] = 1 / 8 / da ** 2
        self.h_RL = (h * da).astype("float32")
        # :69-84 polar pixel coordinates
        isect = np.arange(0, grid_n ** 2)
        cx = cy = grid_n / 2
        i, j = np.unravel_index(isect, (grid_n, grid_n))
        i = i + 1
        j = j + 1
        y = (grid_n + 1 - i - cx - 0.5) * 2 * fov_half / grid_n
        x = (j - cy - 0.5) * 2 * fov_half / grid_n
        r = np.sqrt(x ** 2 + y ** 2)
        phi = np.arctan(y / x)
        phi[x < 0] = phi[x < 0] + np.pi
        phi[phi < 0] = phi[phi < 0] + 2 * np.pi
        self.r = r.reshape(grid_n, grid_n)
        self.phi = phi.reshape(grid_n, grid_n)
        # :104-105 per-detector weight D*cos(nda) (float32: np.cos of a float32 array) and dtheta
        self.weight = (self.D * np.cos(self.nda)).astype("float32")
        self.dtheta = self.theta[1] - self.theta[0]


def _fp(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def weight_sinogram(geo, pj, flip=True):
    """FBP.convert :99-105: flip detector axis, x D cos(gamma), x dtheta -> float32."""
    pj = np.asarray(pj, dtype=np.float32)
    if pj.ndim == 2:
        pj = pj[None]
    if flip:
        pj = np.flip(pj, 2)
    pj = (pj * geo.weight[None, None, :]).astype("float32")
    return np.ascontiguousarray((pj * np.float32(geo.dtheta)).astype("float32"))


def ramp_filter(geo, pjw, f32_accumulate=False):
    """conv_pj :125-131."""
    pjw = np.ascontiguousarray(pjw, dtype=np.float32)
    out = np.zeros_like(pjw)
    fn = _lib().ipdm_oracle_ramp_f32 if f32_accumulate else _lib().ipdm_oracle_ramp
    fn(_fp(pjw, ctypes.c_float), _fp(np.ascontiguousarray(geo.h_RL[:, 0]), ctypes.c_float),
       _fp(out, ctypes.c_float), pjw.shape[0], geo.n_views, geo.n_det)
    return out


def backproject(geo, pj_filtered, pixels=None, want_umap=False):
    """fbp_cpu :166-184 (sequential semantics).  `pixels`: optional int32 flat pixel indices; then
    only those pixels of the returned image are filled."""
    pf = np.ascontiguousarray(pj_filtered, dtype=np.float32)
    bs = pf.shape[0]
    img = np.zeros((bs, geo.grid_n, geo.grid_n), dtype=np.float32)
    phi = np.ascontiguousarray(geo.phi.reshape(-1))
    r = np.ascontiguousarray(geo.r.reshape(-1))
    pix = None if pixels is None else np.ascontiguousarray(pixels, dtype=np.int32)
    npix = 0 if pix is None else pix.size
    umap = None
    if want_umap:
        umap = np.zeros((geo.n_views, npix if pix is not None else geo.grid_n ** 2), dtype=np.float64)
    _lib().ipdm_oracle_backproject(
        _fp(img, ctypes.c_float), bs, _fp(pf, ctypes.c_float), _fp(phi, ctypes.c_double), _fp(r, ctypes.c_double),
        ctypes.c_double(geo.D), geo.grid_n, geo.n_views, geo.n_det, _fp(geo.theta, ctypes.c_double),
        ctypes.c_double(geo.da), ctypes.c_float(float(geo.nda[0])),
        None if pix is None else _fp(pix, ctypes.c_int), npix,
        None if umap is None else _fp(umap, ctypes.c_double))
    return (img, umap) if want_umap else img


def convert(geo, pj, flip=True):
    """FBP.convert :86-122 on numpy input [B, n_views, n_det] -> [B, grid_n, grid_n] float32."""
    pjw = weight_sinogram(geo, pj, flip)
    img = backproject(geo, ramp_filter(geo, pjw))
    if flip:
        img = np.flip(img, 2)
    return np.ascontiguousarray(img)


def convert64(geo, pj, flip=True):
    """float64 ARBITER of convert(): the same function (same float32 constants: detector weights, dtheta, ramp taps,
    nda[0]; same float64 geometry) evaluated in double precision on a float64 sinogram -> float64 image.  Used by the
    tests to measure the float32 oracle and the HIP library against the value both approximate."""
    pj = np.asarray(pj, dtype=np.float64)
    if pj.ndim == 2:
        pj = pj[None]
    if flip:
        pj = np.flip(pj, 2)
    pjw = np.ascontiguousarray(pj * geo.weight[None, None, :].astype(np.float64) * np.float64(np.float32(geo.dtheta)))
    filt = np.zeros_like(pjw)
    _lib().ipdm_oracle_ramp_f64(_fp(pjw, ctypes.c_double), _fp(np.ascontiguousarray(geo.h_RL[:, 0]), ctypes.c_float),
                                _fp(filt, ctypes.c_double), pjw.shape[0], geo.n_views, geo.n_det)
    img = np.zeros((pjw.shape[0], geo.grid_n, geo.grid_n), dtype=np.float64)
    _lib().ipdm_oracle_backproject_f64(
        _fp(img, ctypes.c_double), pjw.shape[0], _fp(filt, ctypes.c_double),
        _fp(np.ascontiguousarray(geo.phi.reshape(-1)), ctypes.c_double), _fp(np.ascontiguousarray(geo.r.reshape(-1)), ctypes.c_double),
        ctypes.c_double(geo.D), geo.grid_n, geo.n_views, geo.n_det, _fp(geo.theta, ctypes.c_double),
        ctypes.c_double(geo.da), ctypes.c_float(float(geo.nda[0])))
    if flip:
        img = np.flip(img, 2)
    return np.ascontiguousarray(img)
