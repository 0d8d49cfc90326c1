"""Whole-dataset evaluation around the sampling path: the npz dataset reader
(Dataset/npz_data_loader.py:55-201), the image-quality metrics of `metric_calculate` (Utils/train_test_utils.py:
792-810), the per-sample / whole-run metric files and result archives (:765-828) and the `test()` / `fit()` drivers
(:274-348).  Host-side Python, as in the reference; the sampling itself is the HIP path of denoiser.py.

Metrics: the reference calls scikit-image 0.19.3 (psnr, ssim), piq 0.8.0 (vif_p, fsim) and its own Utils/NQM.py.  None of
the two packages exists in this image, so psnr / ssim / vif_p restate the published algorithms with those versions'
defaults (the call sites fix win_size=11, data_range=1, chromatic=False); nqm restates Utils/NQM.py and is pinned on
values computed by the reference's own function (tests/golden/metrics.npz).  fsim restates Zhang et al.'s index with
Kovesi's phase congruency in the parameterisation piq 0.8.0 documents -- like psnr / ssim / vif_p it is checked against
its definition, not against the absent package.
"""
import glob
import json
import os

import numpy as np
import torch
from scipy import ndimage


# ----------------------------------------------------------------------------------------------- metrics
def compare_psnr(image_true, image_test, data_range=1):
    """skimage.metrics.peak_signal_noise_ratio: 10 log10(R^2 / mse), the squared error averaged in float64."""
    a, b = np.asarray(image_true), np.asarray(image_test)
    err = np.mean((a - b) ** 2, dtype=np.float64)
    return 10 * np.log10((data_range ** 2) / err)


def compare_ssim(im1, im2, win_size=11, data_range=1, K1=0.01, K2=0.03):
    """skimage.metrics.structural_similarity with its defaults (uniform window, sample covariance, float32 images stay
    float32, border of (win_size-1)//2 cropped, mean in float64)."""
    im1, im2 = np.asarray(im1), np.asarray(im2)
    ft = np.float32 if (im1.dtype == np.float32 and im2.dtype == np.float32) else np.float64
    im1, im2 = im1.astype(ft, copy=False), im2.astype(ft, copy=False)
    npix = win_size ** im1.ndim
    cov_norm = npix / (npix - 1)
    f = lambda x: ndimage.uniform_filter(x, size=win_size)     # noqa: E731
    ux, uy = f(im1), f(im2)
    uxx, uyy, uxy = f(im1 * im1), f(im2 * im2), f(im1 * im2)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return S[tuple(slice(pad, -pad) for _ in range(S.ndim))].mean(dtype=np.float64)


def _gauss_kernel(n, sigma):
    c = np.arange(n, dtype=np.float64) - (n - 1) / 2.0
    g = np.exp(-(c ** 2) / (2 * sigma ** 2))
    k = np.outer(g, g)
    return k / k.sum()


def _valid_conv(x, k):
    n = k.shape[0]
    full = ndimage.correlate(x, k, mode="constant")
    lo = n // 2
    return full[lo:x.shape[0] - (n - 1 - lo), lo:x.shape[1] - (n - 1 - lo)]


def vif_p(x, y, sigma_n_sq=2.0, data_range=1.0):
    """Pixel-domain Visual Information Fidelity (Sheikh & Bovik 2006) in the four-scale form piq.vif_p uses:
    images scaled to [0, 255], Gaussian windows of 17 / 9 / 5 / 3 taps (sigma = taps / 5), valid convolutions, factor-2
    decimation between scales.  x = reference, y = distorted; 2-D arrays (or [1,1,H,W] tensors)."""
    x = np.asarray(x, dtype=np.float64).reshape(np.shape(x)[-2:]) / float(data_range) * 255.0
    y = np.asarray(y, dtype=np.float64).reshape(np.shape(y)[-2:]) / float(data_range) * 255.0
    eps = 1e-8
    num = den = 0.0
    for scale in range(4):
        n = 2 ** (4 - scale) + 1
        k = _gauss_kernel(n, n / 5.0)
        if scale > 0:
            x, y = _valid_conv(x, k)[::2, ::2], _valid_conv(y, k)[::2, ::2]
        mu_x, mu_y = _valid_conv(x, k), _valid_conv(y, k)
        sxx = np.maximum(_valid_conv(x * x, k) - mu_x * mu_x, 0.0)
        syy = np.maximum(_valid_conv(y * y, k) - mu_y * mu_y, 0.0)
        sxy = _valid_conv(x * y, k) - mu_x * mu_y
        g = sxy / (sxx + eps)
        sv = syy - g * sxy
        m = sxx < eps
        g[m], sv[m], sxx[m] = 0.0, syy[m], 0.0
        m = syy < eps
        g[m], sv[m] = 0.0, 0.0
        m = g < 0
        sv[m], g[m] = syy[m], 0.0
        sv = np.maximum(sv, eps)
        num += np.sum(np.log10(1.0 + (g ** 2) * sxx / (sv + sigma_n_sq)))
        den += np.sum(np.log10(1.0 + sxx / sigma_n_sq))
    return (num + eps) / (den + eps)


def _ctf(f):
    return 1.0 / (200 * (2.6 * (0.0192 + 0.114 * f) * np.exp(-(0.114 * f) ** 1.1)))


def NQM(image_origin, image_query, view_angle=1):
    """Noise Quality Measure (Damera-Venkata et al. 2000) as Utils/NQM.py computes it: a 5-band cosine-log pyramid in
    the Fourier domain, contrast masking against the original's band contrasts, global thresholding, then the SNR of the
    restored pair."""
    O, I = np.asarray(image_origin), np.asarray(image_query)
    rows, cols = O.shape
    xp, yp = np.meshgrid(np.arange(-cols / 2, cols / 2), np.arange(-rows / 2, rows / 2))
    r = np.abs(xp + 1j * yp)

    def band(rr, lo, hi, fill, shift):
        inside = (rr >= lo) & (rr <= hi)
        return 0.5 * (1 + np.cos(np.pi * np.log2(rr * inside + fill * (~inside)) - shift))

    filters = [band(r + 2, 1, 4, 4, np.pi), band(r, 1, 4, 4, np.pi), band(r, 2, 8, .5, 0.0), band(r, 4, 16, 4, np.pi),
               band(r, 8, 32, .5, 0.0), band(r, 16, 64, 4, np.pi)]
    FO, FI = np.fft.fft2(O), np.fft.fft2(I)
    bo = [np.real(np.fft.ifft2(np.fft.fftshift(g) * FO)) for g in filters]      # l_0, a_1..a_5
    bi = [np.real(np.fft.ifft2(np.fft.fftshift(g) * FI)) for g in filters]
    y1 = np.zeros_like(bo[0])
    y2 = np.zeros_like(bo[0])
    for k in range(1, 6):
        c = bo[k] / sum(bo[:k])                      # band contrast against everything below it
        ci = bi[k] / sum(bi[:k])
        # contrast masking (cmaskn_modified): where the query's contrast is within the masking threshold of the
        # original's, the query band is replaced by the original band
        ct = _ctf(k)
        cic = np.where(np.abs(ci) > 1, 1.0, ci)
        T = ct * (.86 * ((c / ct) - 1) + .3)
        ai = np.where((np.abs(cic - c) - T) < 0, bo[k], bi[k])
        # global thresholding (gthresh_modified) at the contrast sensitivity of the band's centre frequency
        d = _ctf(2 ** k / view_angle)
        y1 = y1 + np.where(np.abs(c) < d, 0.0, bo[k])
        y2 = y2 + np.where(np.abs(ci) < d, 0.0, ai)
    return 10 * np.log10(np.sum(y1 ** 2) / np.sum((y1 - y2) ** 2))


def _freq_grid(h, w):
    """Normalised frequency coordinates in [-0.5, 0.5), rows x cols ('ij')."""
    def axis(n):
        return np.arange(-(n - 1) / 2, n / 2) / (n - 1) if n % 2 else np.arange(-n / 2, n / 2) / n
    return np.meshgrid(axis(h), axis(w), indexing="ij")


def _phase_congruency(x, scales=4, orientations=4, min_length=6, mult=2, sigma_f=0.55, delta_theta=1.2, k=2.0):
    """Kovesi's phase congruency (the PC_2 measure with his noise compensation) over a bank of `scales` x `orientations`
    log-Gabor filters, as FSIM uses it: per orientation the energy  sum_s (e_s cos(phi) + o_s sin(phi) - |e_s sin(phi) -
    o_s cos(phi)|)  against the mean phase phi, minus a noise threshold estimated from the median response at the finest
    scale, summed over orientations and divided by the summed amplitudes."""
    h, w = x.shape
    eps = np.finfo(x.dtype).eps
    gx, gy = _freq_grid(h, w)
    radius = np.fft.ifftshift(np.sqrt(gx ** 2 + gy ** 2))
    theta = np.fft.ifftshift(np.arctan2(-gy, gx))
    radius[0, 0] = 1
    lowpass = np.fft.ifftshift(1.0 / (1.0 + (np.sqrt(gx ** 2 + gy ** 2) / 0.45) ** (2 * 15)))
    radial = []
    for s in range(scales):
        f0 = 1.0 / (min_length * mult ** s)
        g = np.exp(-(np.log(radius / f0) ** 2) / (2 * np.log(sigma_f) ** 2)) * lowpass
        g[0, 0] = 0
        radial.append(g)
    theta_sigma = np.pi / (orientations * delta_theta)
    fx = np.fft.fft2(x)
    energy_all = np.zeros((h, w), x.dtype)
    an_all = np.zeros((h, w), x.dtype)
    for o in range(orientations):
        ang = o * np.pi / orientations
        ds = np.sin(theta) * np.cos(ang) - np.cos(theta) * np.sin(ang)
        dc = np.cos(theta) * np.cos(ang) + np.sin(theta) * np.sin(ang)
        spread = np.exp(-(np.abs(np.arctan2(ds, dc)) ** 2) / (2 * theta_sigma ** 2))
        filt = [spread * g for g in radial]
        eo = [np.fft.ifft2(fx * f) for f in filt]
        an = sum(np.abs(e) for e in eo)
        sum_e, sum_o = sum(e.real for e in eo), sum(e.imag for e in eo)
        xen = np.sqrt(sum_e ** 2 + sum_o ** 2) + eps
        me, mo = sum_e / xen, sum_o / xen
        energy = sum(e.real * me + e.imag * mo - np.abs(e.real * mo - e.imag * me) for e in eo)
        # noise threshold from the finest scale (Rayleigh model of the noise energy)
        noise_power = (-np.median(np.abs(eo[0]) ** 2) / np.log(0.5)) / np.sum(filt[0] ** 2)
        fi = [np.fft.ifft2(f).real * np.sqrt(h * w) for f in filt]
        sum_an2 = sum(np.sum(f ** 2) for f in fi)
        sum_aiaj = sum(np.sum(fi[a] * fi[b]) for a in range(scales - 1) for b in range(a + 1, scales))
        tau = np.sqrt((2 * noise_power * sum_an2 + 4 * noise_power * sum_aiaj) / 2)
        t = (tau * np.sqrt(np.pi / 2) + k * np.sqrt((2 - np.pi / 2) * tau ** 2)) / 1.7
        energy_all += np.maximum(energy - t, 0)
        an_all += an
    return (energy_all + eps) / (an_all + eps)


def fsim(x, y, data_range=1.0, chromatic=False):
    """Feature Similarity Index (Zhang et al. 2011), luminance form (the call site passes chromatic=False): images on a
    0..255 scale, averaged down to ~256 pixels on the short side, phase-congruency similarity (T1 = 0.85) x Scharr
    gradient-magnitude similarity (T2 = 160), pooled with weights max(PC_x, PC_y).  Restated from the paper and from
    the layout of piq 0.8.0's implementation (package absent here: unpinned)."""
    if chromatic:
        raise NotImplementedError("fsim: only the luminance form used by the sampling harness is built")
    a = np.asarray(x, dtype=np.float32).reshape(np.shape(x)[-2:]) / float(data_range) * 255
    b = np.asarray(y, dtype=np.float32).reshape(np.shape(y)[-2:]) / float(data_range) * 255
    ks = max(1, round(min(a.shape) / 256))
    if ks > 1:
        hh, ww = a.shape[0] // ks * ks, a.shape[1] // ks * ks
        a = a[:hh, :ww].reshape(hh // ks, ks, ww // ks, ks).mean(axis=(1, 3))
        b = b[:hh, :ww].reshape(hh // ks, ks, ww // ks, ks).mean(axis=(1, 3))
    pc_a, pc_b = _phase_congruency(a), _phase_congruency(b)
    scharr = np.array([[-3., 0., 3.], [-10., 0., 10.], [-3., 0., 3.]], dtype=np.float32) / 16

    def grad(img):
        gxx = ndimage.correlate(img, scharr, mode="constant")
        gyy = ndimage.correlate(img, scharr.T, mode="constant")
        return np.sqrt(gxx ** 2 + gyy ** 2)

    ga, gb = grad(a), grad(b)
    s_pc = (2 * pc_a * pc_b + 0.85) / (pc_a ** 2 + pc_b ** 2 + 0.85)
    s_g = (2 * ga * gb + 160) / (ga ** 2 + gb ** 2 + 160)
    pc_max = np.maximum(pc_a, pc_b)
    return float(np.sum(s_g * s_pc * pc_max) / np.sum(pc_max))


# ----------------------------------------------------------------------------------------------- metric bookkeeping
def aggregate_metrics(samples):
    """metric_total_save's arithmetic (Utils/train_test_utils.py:59-118, 812-822): per key the mean over the samples
    that have it and `<key>_std` = population standard deviation, nested dicts preserved."""
    def walk(dicts):
        out = {}
        keys = []
        for d in dicts:
            for k in d:
                if k not in keys:
                    keys.append(k)
        for k in keys:
            vals = [d[k] for d in dicts if k in d]
            if isinstance(vals[0], dict):
                out[k] = walk(vals)
            else:
                out[k] = sum(vals) / len(vals)
        for k in keys:
            vals = [d[k] for d in dicts if k in d]
            if not isinstance(vals[0], dict):
                out[k + "_std"] = (sum((v - out[k]) ** 2 for v in vals) / len(vals)) ** 0.5
        return out
    return walk(list(samples))


# ----------------------------------------------------------------------------------------------- dataset
def _split_path(p):
    """(patient directory, file name) of a dataset file, whichever separator the path was written with (the reference
    splits on a backslash only, Dataset/npz_data_loader.py:119-126)."""
    parts = p.replace("\\", "/").split("/")
    return parts[-2], parts[-1]


class Siemens_dataset_npz(torch.utils.data.Dataset):
    """Dataset/npz_data_loader.py:55-201 for evaluation: four parallel trees `<root>/<patient>/<slice>.npz|npy`
    (low-dose image, full-dose projection, full-dose image, low-dose projection); item = [ld_img, fd_proj, fd_img,
    ld_proj] as [1, H, W] tensors (None for trees not given); projections divided by 10 when proj_clip."""

    def __init__(self, ldproj_path=None, ldimg_path=None, fdproj_path=None, fdimg_path=None, proj_clip=False,
                 img_clip=True, data_type="siemens", patch=None, patch_per_image=None, assign=None):
        if patch is not None:
            raise NotImplementedError("random training patches belong to the training loop (out of scope)")
        self.data_type, self.proj_clip, self.img_clip = data_type, proj_clip, img_clip
        self.patient_name = self.slice_name = None
        self.paths = dict(ldimg=ldimg_path, fdproj=fdproj_path, fdimg=fdimg_path, ldproj=ldproj_path)
        self.files = {}
        for kind in ("fdimg", "fdproj", "ldimg", "ldproj"):          # the reference's order of precedence for names
            root = self.paths[kind]
            if root is None:
                continue
            names = sorted(glob.glob(root + "/*/*"))
            if assign is not None and kind in ("fdimg", "fdproj"):
                names = [n for n in names if _split_path(n)[0] in assign]
            self.files[kind] = names
            if self.patient_name is None:
                self.patient_name = [_split_path(n)[0] for n in names]
                pick = 0 if data_type == "siemens" else -4           # mayo names carry the slice four dots from the end
                self.slice_name = [_split_path(n)[1].split(".")[pick] for n in names]

    @staticmethod
    def get_data(file_path):
        arr = np.load(file_path)
        return arr["arr_0"] if file_path.split(".")[-1] == "npz" else arr

    def _item(self, kind, path):
        a = self.get_data(path)
        if kind in ("fdproj", "ldproj") and self.proj_clip:
            a = a / 10
        return torch.from_numpy(np.ascontiguousarray(a))[None]       # ToTensor() of a 2-D float array

    def __len__(self):
        for kind in ("fdimg", "fdproj", "ldimg", "ldproj"):
            if kind in self.files:
                return len(self.files[kind])
        return 0

    def __getitem__(self, idx):
        return [self._item(k, self.files[k][idx]) if k in self.files else None for k in ("ldimg", "fdproj", "fdimg", "ldproj")]

    def get_data_from_name(self, patient_name, slice_name):
        out = []
        for k in ("ldimg", "fdproj", "fdimg", "ldproj"):
            if k not in self.files:
                out.append(None)
                continue
            hit = [n for n in self.files[k] if patient_name in n and slice_name in n][0]
            out.append(self._item(k, hit))
        return out

    @staticmethod
    def collate(batch):
        return tuple(torch.stack([b[i] for b in batch], 0) if batch[0][i] is not None else None for i in range(4))


# ----------------------------------------------------------------------------------------------- drivers
class EvaluationMixin:
    """The evaluation half of progressive_domain_denoiser (Utils/train_test_utils.py:274-348, 596-828)."""

    METRIC_MODES = ("LDCT", "deProj", "deImg", "deProg", "deProj2img")

    def _init_evaluation(self, save_root):
        from .denoiser import DotDict
        self.metric_each_sample = []
        self.metric_total = DotDict()
        self.metric_clear()
        self.save_root_path = os.path.join(save_root, "save_test_results") if save_root is not None else None
        self.save_path = None
        self.test_dataset = None

    def metric_clear(self):
        from .denoiser import DotDict
        self.metric_instance = DotDict({m: DotDict() for m in self.METRIC_MODES})

    def metric_update(self):
        self.metric_each_sample.append(self.metric_instance)

    def metric_calculate(self, mode="LDCT", **kwargs):
        i, ld = kwargs["it"], kwargs["denoise_result"]
        ld[np.isnan(ld)] = 0.5
        m, want = self.metric_instance[mode], self.opt.metrics
        if "psnr" in want:
            m["psnr_iter_%d" % i] = float(compare_psnr(self.fdct, ld, data_range=1))
        if "ssim" in want:
            m["ssim_iter_%d" % i] = float(compare_ssim(self.fdct, ld, win_size=11, data_range=1))
        if "fsim" in want:
            m["fsim_iter_%d" % i] = float(fsim(self.fdct, ld, data_range=1, chromatic=False))
        if "vif" in want:
            m["vif_iter_%d" % i] = float(vif_p(self.fdct, ld, data_range=1))
        if "nqm" in want:
            m["nqm_iter_%d" % i] = float(NQM(self.fdct, ld))

    def save_path_load(self, epoch, patient_name, slice_name):
        self.save_path = os.path.join(self.save_root_path, "Save_Iter_%s" % epoch, patient_name, slice_name)
        os.makedirs(self.save_path, exist_ok=True)

    def result_data_save(self, data_save=True):
        os.makedirs(self.save_path, exist_ok=True)
        if data_save:
            for ftype, fdata in (("prog_denoise_result", self.progressive_denoise_result),
                                 ("proj_denoise_result", self.proj_denoise_result),
                                 ("img_denoise_result", self.img_denoise_result),
                                 ("proj_denoise_result_2img", self.proj_denoise_convert2img_result)):
                if len(fdata) > 0:
                    np.savez_compressed(os.path.join(self.save_path, ftype + ".npz"), **fdata)
        with open(os.path.join(self.save_path, "metric.json"), "w") as f:
            f.write(json.dumps(self.metric_instance, sort_keys=False, indent=4, separators=(",", ": ")))

    def metric_total_save(self, epoch):
        from .denoiser import DotDict
        self.metric_total = DotDict(aggregate_metrics(self.metric_each_sample))
        out = os.path.join(self.save_root_path, "Save_Iter_%s" % epoch)
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "metric.json"), "w") as f:
            f.write(json.dumps(self.metric_total, sort_keys=False, indent=4, separators=(",", ": ")))

    # ---- figures (Utils/train_test_utils.py:596-763).  Drawn with matplotlib when it is importable (it is in the
    # image); a box without it still gets every metric and a warning instead of an exception.
    _WINDOW = ((-160 + 1024) / 4096, (240 + 1024) / 4096)          # the reference's display window [-160, 240] HU

    @staticmethod
    def _pyplot():
        try:
            import matplotlib
            if not os.environ.get("DISPLAY") and "matplotlib.pyplot" not in __import__("sys").modules:
                matplotlib.use("Agg")
            from matplotlib import pyplot as plt
            return plt
        except Exception as e:                                    # noqa: BLE001 - any import problem means "no figures"
            import warnings
            warnings.warn("result_figure_save: matplotlib unavailable (%s); metrics computed, figures not drawn" % e)
            return None

    def _panel(self, ax, title, image, caption=None, cap_y=-0.15):
        ax.set_title(title, fontsize=35, y=1.02)
        if caption is not None:
            ax.text(x=0.5, y=cap_y, s=caption, fontsize=25, horizontalalignment="center", transform=ax.transAxes)
        ax.set_xticks([])
        ax.set_yticks([])
        ax.imshow(image, "gray", vmin=self._WINDOW[0], vmax=self._WINDOW[1])

    def _caption(self, mode, it):
        m = self.metric_instance[mode]
        parts = ["%s=%.2f" % (k.upper(), m["%s_iter_%d" % (k, it)]) for k in ("psnr", "ssim") if "%s_iter_%d" % (k, it) in m]
        return " , ".join(parts) if parts else None

    def result_figure_save(self, mode="progressive", display=True, only_metric=False):
        """Metrics of every stored iterate of `mode`'s result dict against the full-dose image, in the reference's
        order (LDCT first; `progressive` also scores the converted proj-domain iterates as `deProj`), and -- unless
        only_metric -- the reference's figure (`progressive.png` / `deImg.png` / `deProj2img.png` / `dProj.png` under
        self.save_path).  Returns -1 on an unknown mode, like the reference."""
        from .denoiser import miu2pixel
        if mode not in ("progressive", "dimg", "dproj", "dproj2img"):
            print('ValueError:mode should be one of: "progressive","dimg","dproj","dproj2img"')
            return -1
        plt = None if (only_metric and mode != "dproj") else self._pyplot()
        draw = plt is not None and not only_metric
        if draw and self.save_path is None:
            import warnings
            warnings.warn("result_figure_save: no save path (call save_path_load first); figure not written")
        fig = None

        def savefig(name, dpi):
            if self.save_path is not None:
                os.makedirs(self.save_path, exist_ok=True)
                plt.savefig(os.path.join(self.save_path, name), dpi=dpi)

        if mode == "dproj":                                        # residual maps against the full-dose sinogram; no metrics
            if plt is None or self.fdproj is None:
                return None
            target = np.abs(np.asarray(self.fdproj) - self.ldproj_np)
            n = len(self.proj_denoise_result)
            fig, ax = plt.subplots(1, 1 + n, figsize=(30, 30), squeeze=False)
            lo, hi = target.min(), target.max()
            res = [target] + [np.abs(self.proj_denoise_result["iter_%d" % i][0, 0] - np.asarray(self.fdproj)) for i in range(1, n + 1)]
            for k, r in enumerate(res):
                ax[0, k].set_title("res target" if k == 0 else "deProj iter%d" % k, fontsize=35, y=1.02)
                ax[0, k].set_xticks([])
                ax[0, k].set_yticks([])
                ax[0, k].imshow(r, "inferno", vmin=lo, vmax=hi)
            savefig("dProj.png", 100)
        else:
            self.metric_calculate(mode="LDCT", it=0, denoise_result=self.ldct_np)
            if mode == "progressive":
                top, store, key = self.proj_denoise_convert2img_result, self.progressive_denoise_result, "deProg"
                cols = 1 + max(len(store), len(top))
                if draw:
                    fig, ax = plt.subplots(2, cols, figsize=(7 * cols, 16), squeeze=False)
                    self._panel(ax[0, 0], "LDCT", self.ldct_np, self._caption("LDCT", 0), -0.09)
                for i in range(1, len(top) + 1):
                    r = miu2pixel(top["iter_%d" % i][0, 0])
                    self.metric_calculate(mode="deProj", it=i, denoise_result=r)
                    if draw:
                        self._panel(ax[0, i], "Proj iter%d" % i, r, self._caption("deProj", i), -0.09)
                for i in range(1, len(store) + 1):
                    it = len(store) + 1 - i
                    r = miu2pixel(store["iter_%d" % it][0, 0])
                    self.metric_calculate(mode=key, it=it, denoise_result=r)
                    if draw:
                        self._panel(ax[1, i], "Img iter%d" % it, r, self._caption(key, it), -0.09)
                if draw:
                    self._panel(ax[1, 0], "FDCT", self.fdct)
                    savefig("progressive.png", 100)
            else:
                store, key, label, fname = {"dimg": (self.img_denoise_result, "deImg", "Img", "deImg.png"),
                                            "dproj2img": (self.proj_denoise_convert2img_result, "deProj2img", "Proj",
                                                          "deProj2img.png")}[mode]
                n = len(store)
                if draw:
                    fig, ax = plt.subplots(1, 2 + n, figsize=(7 * (2 + n), 7), squeeze=False)
                    self._panel(ax[0, 0], "LDCT", self.ldct_np, self._caption("LDCT", 0))
                    self._panel(ax[0, 1], "FDCT", self.fdct)
                for i in range(1, n + 1):
                    it = n + 1 - i
                    r = miu2pixel(store["iter_%d" % it][0, 0])
                    self.metric_calculate(mode=key, it=it, denoise_result=r)
                    if draw:
                        self._panel(ax[0, i + 1], "%s iter%d" % (label, it), r, self._caption(key, it))
                if draw:
                    savefig(fname, 200)
        if fig is not None and not display:
            plt.close(fig)
        return None

    def init_data_loader(self):
        o = self.opt
        if "train" in o.mode:
            raise NotImplementedError("training is out of scope of this build (DESIGN.md section 7)")
        self.test_dataset = Siemens_dataset_npz(ldimg_path=o.test_dataset_path_LD_img, fdimg_path=o.test_dataset_path_FD_img,
                                                ldproj_path=o.test_dataset_path_LD_proj,
                                                fdproj_path=o.test_dataset_path_FD_proj, proj_clip=o.clip_proj,
                                                img_clip=o.clip_img, data_type=o.data_type)

    @torch.no_grad()
    def test(self, epoch):
        o = self.opt
        if self.test_dataset is None:
            self.init_data_loader()
        if o.test_numbers <= 0:
            o.test_numbers = len(self.test_dataset)
        np.random.seed(9527)
        ids = np.sort(np.random.choice(len(self.test_dataset), o.test_numbers, replace=False))
        for idx in range(o.test_numbers):
            ld_img, fd_proj, fd_img, ld_proj = self.test_dataset[ids[idx]]
            ld_img, fd_img = ld_img[None], fd_img[None]
            ld_proj = ld_proj[None] if ld_proj is not None else None
            self.temp_clear()
            self.metric_clear()
            self.save_path_load(epoch, self.test_dataset.patient_name[ids[idx]], self.test_dataset.slice_name[ids[idx]])
            self.data_sample_load(ldct=ld_img, ldproj=ld_proj, fdproj=fd_proj, fdct=fd_img)
            if o.mode in ("train_proj", "test_proj"):
                self.proj_denoiser(self.ldproj)
                self.result_figure_save(mode="dproj2img", display=False, only_metric=not o.display_result)
            if o.mode in ("train_img", "test_img"):
                self.img_denoiser(self.ldct, mode="img_only")
                self.result_figure_save(mode="dimg", display=False, only_metric=not o.display_result)
            if o.mode == "test_prog":
                self.progressive_denoiser()
                self.result_figure_save(mode="progressive", display=False, only_metric=not o.display_result)
            self.result_data_save(data_save=o.test_result_data_save)
            self.metric_update()
        self.metric_total_save(epoch)

    def fit(self):
        if "test" in self.opt.mode:
            return self.test(0)
        raise NotImplementedError("training is out of scope of this build (DESIGN.md section 7)")
