"""CPU tests of the evaluation row: metrics, metric aggregation, dataset reader, result writers."""
import json
import os
import types

import numpy as np
import torch

from ipdm_pytorch_amd import evaluate as ev, synth
from ipdm_pytorch_amd.denoiser import DotDict, ResultTempDict


def _pair(n, seed, noise):
    yy, xx = np.mgrid[0:n, 0:n] / float(n)
    ref = (0.5 + 0.3 * np.sin(9 * xx) * np.cos(7 * yy) + 0.15 * (((xx - .4) ** 2 + (yy - .6) ** 2) < .04)).astype(np.float32)
    return ref, (ref + noise * synth.hash_normal((n, n), seed)).astype(np.float32)


def test_nqm_matches_reference_values(golden):
    g = golden("metrics")
    for tag, (n, seed, noise) in {"a": (128, 81, 0.02), "b": (96, 82, 0.08), "c": (512, 83, 0.01)}.items():
        ref, qry = _pair(n, seed, noise)
        assert abs(ev.NQM(ref, qry) - float(g["nqm_" + tag])) <= 1e-9 * abs(float(g["nqm_" + tag]))


def test_psnr_and_ssim_definitions():
    ref, qry = _pair(40, 5, 0.05)
    mse = np.mean((ref.astype(np.float64) - qry.astype(np.float64)) ** 2)
    assert abs(ev.compare_psnr(ref, qry, data_range=1) - 10 * np.log10(1 / mse)) < 1e-5
    # SSIM against the textbook per-window form (sample covariance, 11x11 uniform windows fully inside the image)
    a, b = ref.astype(np.float64), qry.astype(np.float64)
    C1, C2, vals = 0.01 ** 2, 0.03 ** 2, []
    for i in range(5, 35):
        for j in range(5, 35):
            wa, wb = a[i - 5:i + 6, j - 5:j + 6].ravel(), b[i - 5:i + 6, j - 5:j + 6].ravel()
            cov = np.cov(wa, wb, ddof=1)
            vals.append((2 * wa.mean() * wb.mean() + C1) * (2 * cov[0, 1] + C2) /
                        ((wa.mean() ** 2 + wb.mean() ** 2 + C1) * (cov[0, 0] + cov[1, 1] + C2)))
    assert abs(ev.compare_ssim(a, b, win_size=11, data_range=1) - np.mean(vals)) < 1e-10
    assert abs(ev.compare_ssim(ref, qry, win_size=11, data_range=1) - np.mean(vals)) < 5e-5      # float32 path
    assert ev.compare_ssim(ref, ref) == 1.0


def test_vif_behaviour():
    ref, q1 = _pair(128, 7, 0.02)
    _, q2 = _pair(128, 7, 0.1)
    assert abs(ev.vif_p(ref, ref) - 1.0) < 1e-6
    v1, v2 = ev.vif_p(ref, q1), ev.vif_p(ref, q2)
    assert 0 < v2 < v1 < 1
    assert abs(ev.vif_p(torch.from_numpy(ref)[None, None], torch.from_numpy(q1)[None, None], data_range=1) - v1) < 1e-12


def test_aggregate_metrics_mean_and_population_std():
    s = [dict(LDCT=dict(psnr_iter_0=30.0, ssim_iter_0=0.8), deProg=dict(psnr_iter_1=35.0)),
         dict(LDCT=dict(psnr_iter_0=32.0, ssim_iter_0=0.9), deProg=dict(psnr_iter_1=36.0, psnr_iter_2=40.0)),
         dict(LDCT=dict(psnr_iter_0=34.0, ssim_iter_0=1.0), deProg=dict())]
    t = ev.aggregate_metrics(s)
    assert t["LDCT"]["psnr_iter_0"] == 32.0 and abs(t["LDCT"]["psnr_iter_0_std"] - np.std([30, 32, 34])) < 1e-12
    assert t["deProg"]["psnr_iter_1"] == 35.5 and t["deProg"]["psnr_iter_2"] == 40.0 and t["deProg"]["psnr_iter_2_std"] == 0.0
    assert list(t["LDCT"]) == ["psnr_iter_0", "ssim_iter_0", "psnr_iter_0_std", "ssim_iter_0_std"]


def _write_tree(root, kind, names, shape, seed):
    for i, (pat, sl) in enumerate(names):
        os.makedirs(os.path.join(root, kind, pat), exist_ok=True)
        np.savez(os.path.join(root, kind, pat, sl + ".npz"), synth.hash_uniform(shape, seed + i).astype(np.float32))


def test_dataset_reader(tmp_path):
    root = str(tmp_path)
    names = [("L067", "slice_003"), ("L067", "slice_001"), ("L096", "slice_010")]
    _write_tree(root, "fdimg", names, (16, 16), 1)
    _write_tree(root, "ldimg", names, (16, 16), 11)
    _write_tree(root, "ldproj", names, (20, 12), 21)
    ds = ev.Siemens_dataset_npz(ldimg_path=root + "/ldimg", fdimg_path=root + "/fdimg", ldproj_path=root + "/ldproj", proj_clip=True)
    assert len(ds) == 3
    assert ds.patient_name == ["L067", "L067", "L096"] and ds.slice_name == ["slice_001", "slice_003", "slice_010"]
    ld_img, fd_proj, fd_img, ld_proj = ds[0]
    assert fd_proj is None and tuple(ld_img.shape) == (1, 16, 16) and tuple(ld_proj.shape) == (1, 20, 12)
    assert np.array_equal(fd_img[0].numpy(), synth.hash_uniform((16, 16), 2).astype(np.float32))      # sorted: slice_001 first
    assert np.allclose(ld_proj[0].numpy(), synth.hash_uniform((20, 12), 22).astype(np.float32) / 10)   # proj_clip
    assert ev._split_path("D:\\data\\L067\\slice_003.npz") == ("L067", "slice_003.npz")           # the reference's paths
    batch = ds.collate([ds[0], ds[2]])
    assert batch[1] is None and tuple(batch[0].shape) == (2, 1, 16, 16)
    got = ds.get_data_from_name("L096", "slice_010")
    assert torch.equal(got[2], ds[2][2])


def test_result_writers_and_metric_calculate(tmp_path):
    den = types.SimpleNamespace(opt=types.SimpleNamespace(metrics=["psnr", "ssim", "vif", "nqm"]))
    for name in ("metric_clear", "metric_calculate", "metric_update", "save_path_load", "result_data_save", "metric_total_save",
                 "result_figure_save", "_init_evaluation"):
        setattr(den, name, types.MethodType(getattr(ev.EvaluationMixin, name), den))
    den.METRIC_MODES = ev.EvaluationMixin.METRIC_MODES
    den._init_evaluation(str(tmp_path / "IPDM_default"))
    ref, qry = _pair(64, 3, 0.03)
    den.fdct, den.ldct_np = ref, qry
    den.progressive_denoise_result = ResultTempDict()
    den.proj_denoise_result, den.img_denoise_result, den.proj_denoise_convert2img_result = ResultTempDict(), ResultTempDict(), ResultTempDict()
    # two stored iterates in mu units: the identity and a noisier image
    from oracle import diffusion as od
    mu = 0.183 * (1 + (ref * 4096 - 1024 + 24) / 1000.0)
    den.progressive_denoise_result["iter_1"] = (mu + 0.001 * synth.hash_normal((64, 64), 9)).astype(np.float32)[None, None]
    den.progressive_denoise_result["iter_2"] = mu.astype(np.float32)[None, None]
    for k in range(2):
        den.metric_clear()
        den.save_path_load(0, "L067", "slice_%03d" % k)
        assert den.result_figure_save(mode="progressive", display=False, only_metric=True) is None
        den.result_data_save(data_save=True)
        den.metric_update()
    m = den.metric_instance
    assert set(m["LDCT"]) == {"psnr_iter_0", "ssim_iter_0", "vif_iter_0", "nqm_iter_0"}
    assert m["deProg"]["psnr_iter_2"] > m["deProg"]["psnr_iter_1"] > m["LDCT"]["psnr_iter_0"]
    assert abs(m["LDCT"]["psnr_iter_0"] - ev.compare_psnr(ref, qry)) < 1e-12
    assert np.abs(od.miu2pixel(torch.from_numpy(mu.astype(np.float32))).numpy() - ref).max() < 1e-5
    saved = os.path.join(den.save_root_path, "Save_Iter_0", "L067", "slice_001")
    assert os.path.isfile(os.path.join(saved, "prog_denoise_result.npz")) and not os.path.exists(os.path.join(saved, "img_denoise_result.npz"))
    assert json.load(open(os.path.join(saved, "metric.json")))["deProg"]["ssim_iter_2"] > 0.999
    den.metric_total_save(0)
    tot = json.load(open(os.path.join(den.save_root_path, "Save_Iter_0", "metric.json")))
    assert tot["LDCT"]["psnr_iter_0_std"] == 0.0 and abs(tot["LDCT"]["psnr_iter_0"] - m["LDCT"]["psnr_iter_0"]) < 1e-12
    den.opt.metrics = ["fsim"]
    den.metric_calculate(mode="LDCT", it=0, denoise_result=qry.copy())
    assert 0 < den.metric_instance["LDCT"]["fsim_iter_0"] <= 1


def test_result_figure_save_default_arguments(tmp_path):
    """The notebook's cell 2 calls result_figure_save(mode="progressive") with its defaults (test_sample.ipynb cell 2;
    Utils/train_test_utils.py:596): metrics in the reference's order (progressive also scores the converted proj iterates
    as deProj) and the figure file the reference writes; every mode, never an exception."""
    den = types.SimpleNamespace(opt=types.SimpleNamespace(metrics=["psnr", "ssim"]))
    for name in ("metric_clear", "metric_calculate", "metric_update", "save_path_load", "result_figure_save", "_init_evaluation",
                 "_panel", "_caption"):
        setattr(den, name, types.MethodType(getattr(ev.EvaluationMixin, name), den))
    den._pyplot, den._WINDOW = ev.EvaluationMixin._pyplot, ev.EvaluationMixin._WINDOW
    den.METRIC_MODES = ev.EvaluationMixin.METRIC_MODES
    den._init_evaluation(str(tmp_path / "figs"))
    ref, qry = _pair(48, 5, 0.03)
    mu = (0.183 * (1 + (ref * 4096 - 1024 + 24) / 1000.0)).astype(np.float32)[None, None]
    den.fdct, den.ldct_np = ref, qry
    den.fdproj = synth.hash_uniform((20, 12), 3).astype(np.float32)
    den.ldproj_np = den.fdproj + 0.01
    stores = {}
    for name, n in (("progressive_denoise_result", 2), ("proj_denoise_convert2img_result", 3), ("img_denoise_result", 1),
                    ("proj_denoise_result", 2)):
        d = ResultTempDict()
        for i in range(1, n + 1):
            d["iter_%d" % i] = (den.fdproj[None, None] + 0.001 * i) if name == "proj_denoise_result" else mu + 1e-4 * i
        setattr(den, name, d)
        stores[name] = d
    den.save_path_load(0, "L067", "slice_000")
    for mode, fname in (("progressive", "progressive.png"), ("dimg", "deImg.png"), ("dproj2img", "deProj2img.png"),
                        ("dproj", "dProj.png")):
        den.metric_clear()
        assert den.result_figure_save(mode=mode, display=False) is None
        assert os.path.getsize(os.path.join(den.save_path, fname)) > 1000
    den.metric_clear()
    assert den.result_figure_save(mode="progressive") is None                       # the notebook's literal call
    m = den.metric_instance
    assert list(m["deProj"]) == ["psnr_iter_1", "ssim_iter_1", "psnr_iter_2", "ssim_iter_2", "psnr_iter_3", "ssim_iter_3"]
    assert list(m["deProg"]) == ["psnr_iter_2", "ssim_iter_2", "psnr_iter_1", "ssim_iter_1"]       # last iterate first
    assert den.result_figure_save(mode="nonsense") == -1
    import matplotlib.pyplot as plt
    plt.close("all")


def test_fsim_behaviour():
    ref, q1 = _pair(256, 7, 0.02)
    _, q2 = _pair(256, 7, 0.1)
    assert abs(ev.fsim(ref, ref) - 1.0) < 1e-6
    f1, f2 = ev.fsim(ref, q1), ev.fsim(ref, q2)
    assert 0 < f2 < f1 < 1
    big, bigq = _pair(512, 9, 0.03)                          # 512 -> averaged down by 2 before the analysis
    assert 0 < ev.fsim(big, bigq) < 1
    pc = ev._phase_congruency(ref * 255)
    assert pc.shape == ref.shape and pc.min() >= 0 and pc.max() <= 1 + 1e-6


def test_yeo_johnson_round_trip_and_per_slice():
    """opt.normal helpers (Model/model.py:762-808): sklearn's standardised Yeo-Johnson transform, one per slice."""
    from sklearn.preprocessing import PowerTransformer
    from ipdm_pytorch_amd.normalize import yeo_johnson_transform, yeo_johnson_inverse_transform
    x = torch.from_numpy((synth.hash_uniform((2, 1, 24, 20), 77) ** 2 * 3).astype(np.float32))
    y, trs = yeo_johnson_transform(x)
    assert tuple(y.shape) == tuple(x.shape) and len(trs) == 2
    for b in range(2):                                   # a batch of B equals B single calls, and equals bare sklearn
        want = PowerTransformer(method="yeo-johnson").fit_transform(x[b].numpy().reshape(-1, 1)).reshape(1, 24, 20)
        assert np.array_equal(y[b].numpy(), want)
        assert abs(float(y[b].mean())) < 1e-6 and abs(float(y[b].std(unbiased=False)) - 1) < 1e-6
    back = yeo_johnson_inverse_transform(y, trs)
    assert np.abs(back.numpy() - x.numpy()).max() < 1e-5
    assert np.array_equal(yeo_johnson_inverse_transform(y[:1], trs[0]).numpy(), back[:1].numpy())     # bare transformer form
