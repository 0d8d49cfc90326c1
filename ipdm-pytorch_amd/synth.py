"""Deterministic synthetic inputs and weights (library-independent integer hashing).

There are no pretrained weights or Mayo slices in the reference tree, so parity
tests, smoke() and bench.py use: (i) SplitMix64-hashed uniform/normal arrays -- a pure function of
(seed, index), identical on every machine and numpy version; (ii) reference-layout state_dicts with
fan-in-scaled weights (activations stay O(1)); (iii) an analytic ellipse phantom with exact fan-beam
line integrals in the FBP geometry and the reference's low-dose noise model
(Utils/Low_dose_CT_simulate.py:38-44).
"""
import math

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def hash_uniform(shape, seed):
    """float32 uniform in (0,1): exact 24-bit mantissas from SplitMix64(seed, index)."""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(12345))
        bits = _splitmix64(idx ^ key)
    u = ((bits >> np.uint64(40)).astype(np.float64) + 0.5) / 16777216.0
    return u.astype(np.float32).reshape(shape)


def hash_normal(shape, seed):
    """float32 N(0,1) by Box-Muller over two hashed uniforms (float64 math, rounded once)."""
    u1 = hash_uniform(shape, 2 * seed + 1).astype(np.float64)
    u2 = hash_uniform(shape, 2 * seed + 2).astype(np.float64)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * math.pi * u2)).astype(np.float32)


def synth_state_dict(param_shapes, seed=0, affine_jitter=0.1):
    """Reference-layout state_dict {name: float32 ndarray} for the given ordered {name: shape}.

    conv/linear weights ~ U(-b, b), b = sqrt(3/fan_in)  (unit-gain: variance 1/fan_in);
    biases ~ U(-0.1, 0.1); GroupNorm gamma = 1 + jitter*U(-1,1), beta = jitter*U(-1,1)."""
    sd = {}
    for k, (name, shape) in enumerate(param_shapes.items()):
        u = hash_uniform(shape, seed * 100003 + k) * 2.0 - 1.0
        is_norm = (".conv1.0." in name or ".conv2.0." in name or ".norm." in name or name.startswith("out.0."))
        if is_norm:
            w = (1.0 + affine_jitter * u) if name.endswith("weight") else affine_jitter * u
        elif name.endswith("bias"):
            w = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            w = u * math.sqrt(3.0 / fan_in)
        sd[name] = np.ascontiguousarray(w, dtype=np.float32)
    return sd


# ------------------------------------------------------------------ phantom + sinogram
def ellipse_phantom(slice_id):
    """List of (x0, y0, a, b, angle, mu) ellipses (cm, additive attenuation 1/cm): body, organs,
    bone ring (outer + inner), air pockets."""
    u = hash_uniform((64,), 7919 + slice_id).astype(np.float64)
    ell = [(0.0, 0.0, 15.0 + 2 * u[0], 11.0 + 2 * u[1], 0.0, 0.19)]
    k = 2
    for _ in range(7):
        x0 = (u[k] - 0.5) * 16
        y0 = (u[k + 1] - 0.5) * 11
        a = 1.0 + 2.5 * u[k + 2]
        b = 1.0 + 2.5 * u[k + 3]
        ang = math.pi * u[k + 4]
        mu = (u[k + 5] - 0.5) * 0.08
        ell.append((x0, y0, a, b, ang, mu))
        k += 6
    ell.append((0.0, -7.0, 2.2, 2.2, 0.0, 0.26))      # vertebra (bone ~0.45 total)
    ell.append((0.0, -7.0, 1.0, 1.0, 0.0, -0.24))     # canal
    ell.append((-6.0 + 2 * u[50], 2.0, 3.5, 4.5, 0.3, -0.15))   # lung-ish
    ell.append((6.0 + 2 * u[51], 2.0, 3.5, 4.5, -0.3, -0.15))
    return ell


def rasterize(ell, grid_n=512, fov_half=21.0):
    """mu image on the FBP pixel grid (Recon/FBP_kernel.py:69-84 pixel-centre convention)."""
    i = np.arange(1, grid_n + 1)[:, None]
    j = np.arange(1, grid_n + 1)[None, :]
    y = (grid_n + 1 - i - grid_n / 2 - 0.5) * 2 * fov_half / grid_n
    x = (j - grid_n / 2 - 0.5) * 2 * fov_half / grid_n
    img = np.zeros((grid_n, grid_n), dtype=np.float64)
    for (x0, y0, a, b, ang, mu) in ell:
        ca, sa = math.cos(ang), math.sin(ang)
        xr = (x - x0) * ca + (y - y0) * sa
        yr = -(x - x0) * sa + (y - y0) * ca
        img += mu * ((xr / a) ** 2 + (yr / b) ** 2 <= 1.0)
    return img.astype(np.float32)


def fan_sinogram(ell, n_views=2000, n_det=912, da=0.0010125, det_offset=3.75, dtheta_deg=0.18, D=59.5):
    """Exact line integrals of the ellipses along the rays of the equiangular fan-beam geometry of
    Recon/FBP_kernel.py (source at angle beta on radius D, ray at fan angle gamma), in the
    reference's stored orientation (detector axis flipped w.r.t. FBP's internal one, cf. :100)."""
    theta = (np.arange(n_views) * dtheta_deg) / 180 * np.pi
    start = (-n_det / 2 + 0.5 + det_offset) * da
    gam = start + np.arange(n_det) * da
    beta = theta - np.pi / 2
    # source position and ray direction, consistent with alpha = atan(r sin(th)/(D + r cos(th))), th = theta+phi
    # i.e. in the frame rotated by -theta the source sits at (-D, 0) looking along +x.
    sino = np.zeros((n_views, n_det), dtype=np.float64)
    ct, st = np.cos(theta)[:, None], np.sin(theta)[:, None]
    cg, sg = np.cos(gam)[None, :], np.sin(gam)[None, :]
    # rotated-frame ray: p(s) = (-D, 0) + s (cos g, sin g); world = R(-theta) p  (x' = x cos th - y sin th ... inverse)
    sx = -D * ct            # world source: rotate (-D,0) by -theta: (x cos t + y sin t, -x sin t + y cos t)
    sy = D * st
    dx = cg * ct + sg * st
    dy = -cg * st + sg * ct
    del beta
    for (x0, y0, a, b, ang, mu) in ell:
        # the reference's stored sinograms reconstruct x-mirrored w.r.t. this ray parametrisation
        # (FBP.convert flips detector and image axes, Recon/FBP_kernel.py:100,118): project the mirror image
        x0, ang = -x0, -ang
        ca, sa = math.cos(ang), math.sin(ang)
        px = ((sx - x0) * ca + (sy - y0) * sa) / a
        py = (-(sx - x0) * sa + (sy - y0) * ca) / b
        qx = (dx * ca + dy * sa) / a
        qy = (-dx * sa + dy * ca) / b
        A = qx * qx + qy * qy
        Bq = px * qx + py * qy
        Cq = px * px + py * py - 1.0
        disc = Bq * Bq - A * Cq
        sino += mu * np.where(disc > 0, 2.0 * np.sqrt(np.maximum(disc, 0)) / A, 0.0)
    return np.ascontiguousarray(sino[:, ::-1]).astype(np.float32)


def low_dose(sino, seed, factor=0.25, n0=1.4e5, ne=5.8):
    """Utils/Low_dose_CT_simulate.py:38-44 noise model with hashed normals instead of np.random:
    transmitted counts N = N0*factor*exp(-p) + sqrt(that)*z1 + sqrt(Ne)*z2; p_ld = -log(N/(N0*factor))."""
    z1 = hash_normal(sino.shape, 31 + 2 * seed).astype(np.float64)
    z2 = hash_normal(sino.shape, 32 + 2 * seed).astype(np.float64)
    lam = n0 * factor * np.exp(-sino.astype(np.float64))
    n = lam + np.sqrt(lam) * z1 + math.sqrt(ne) * z2
    n = np.maximum(n, 1.0)
    return (-np.log(n / (n0 * factor))).astype(np.float32)
