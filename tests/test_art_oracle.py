"""CPU tests of the ART row: the two data tables against samples of the reference's files, and the
oracle's restatement of the SART / projector arithmetic through properties (it is "parity unpinned": the CUDA reference
cannot run in this image, see oracle/art_oracle.c)."""
import numpy as np

from ipdm_pytorch_amd import art
from oracle import art as oa


def test_area_table_and_view_angles_match_reference_files(golden):
    g = golden("art_tables")
    lut = art.area_lut()
    assert lut.shape == tuple(g["lut_shape"]) and lut.dtype == np.float32
    got = lut[np.ix_(g["rows"], g["cols"])]
    assert np.abs(got - g["lut"]).max() <= 2e-17                     # float32-identical up to one denormal-scale entry
    assert abs(float(lut.astype(np.float64).sum()) - float(g["lut_sum"])) <= 1e-12 * float(g["lut_sum"])
    th = art.view_angles()
    assert th.size == int(g["theta_n"]) and np.array_equal(th[g["theta_idx"]], g["theta"])
    assert np.array_equal(oa.area_lut(np.float32(42.0) / np.float32(512.0)), lut) and np.array_equal(oa.view_angles(), th)


def _small(nx=48, nr=96, na=72):
    kw = dict(nx=nx, nr=nr, na=na, dr=0.0010125 * 912 / nr, offset_r=-3.75 * nr / 912)
    go = oa.geometry(**kw)
    return go, art.area_lut(go.dx), art.view_angles(na, 360.0 / na)


def test_oracle_projector_line_integrals_and_linearity():
    go, lut, betas = _small()
    nx = go.nx
    yy, xx = np.mgrid[0:nx, 0:nx]
    disk = (((xx - nx / 2 + 0.5) ** 2 + (yy - nx / 2 + 0.5) ** 2) < (nx * 0.3) ** 2).astype(np.float32)
    box = np.zeros((nx, nx), np.float32)
    box[10:20, 25:33] = 0.5
    p = oa.project(go, lut, betas, np.stack([disk, box, 2 * disk + box]))
    # a centred disk of unit attenuation: every view sees the same profile, peak = its diameter (in cm)
    peak = p[0].max(axis=1)
    # (to within the rasterised edge: one 0.875 cm pixel on either end of the 25 cm chord)
    assert np.abs(peak - peak.mean()).max() <= 1.5 * go.dx
    assert abs(peak.mean() - 2 * nx * 0.3 * go.dx) <= 0.5 * go.dx
    assert np.abs(p[2] - (2 * p[0] + p[1])).max() <= 1e-5 * p[2].max()


def test_oracle_sart_converges_and_tv_is_finite():
    go, lut, betas = _small()
    nx = go.nx
    yy, xx = np.mgrid[0:nx, 0:nx]
    vol = (((xx - 22) ** 2 + (yy - 26) ** 2) < 14 ** 2).astype(np.float32) * 0.2
    vol[14:20, 24:30] += 0.1
    p = oa.project(go, lut, betas, vol[None])
    errs = [np.sqrt(((oa.reconstruct(go, lut, betas, p, n, 0, permute=False)[0] - vol) ** 2).mean()) for n in (1, 3, 6)]
    assert errs[0] > errs[1] > errs[2] and errs[2] <= 0.05 * vol.max()
    r = oa.reconstruct(go, lut, betas, p, 3, 2, permute=True)
    assert r.shape == (1, nx, nx) and np.isfinite(r).all()
    assert np.array_equal(oa.reconstruct(go, lut, betas, p, 0, 0), np.zeros((1, nx, nx), np.float32))


def test_oracle_projector_against_analytic_ellipse_integrals():
    """A check that does not share the restatement's reading of the CUDA code: at the reference's full geometry the
    projection of a RASTERISED ellipse phantom must equal the EXACT fan-beam line integrals of the ellipses
    (synth.fan_sinogram -- the sinograms the reference's own FBP.convert reconstructs back to the phantom, fbp.npz) up to
    the pixelisation of the edges: 0.29 % relative rms, median 5e-4 of values up to 6.4 (measured here)."""
    import ipdm_pytorch_amd  # noqa: F401
    from ipdm_pytorch_amd import synth
    g = oa.geometry()
    ell = synth.ellipse_phantom(3)
    mu = synth.rasterize(ell).astype(np.float32)
    ana = synth.fan_sinogram(ell)
    p = oa.project(g, art.area_lut(), art.view_angles(), mu.T.copy()[None])[0]      # proj_torch sees the volume x-major
    d = np.abs(p - ana)
    assert np.sqrt((d ** 2).mean()) <= 5e-3 * np.sqrt((ana ** 2).mean())
    assert np.median(d) <= 1.5e-3 and np.percentile(d, 99) <= 0.06
