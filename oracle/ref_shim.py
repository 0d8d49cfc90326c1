"""Import shim for the *reference* (LFY1998/IPDM-PyTorch at /root/reference).

TEST INFRASTRUCTURE ONLY.  This module exists so that the golden-vector generator
(tests/golden/make_golden.py) and the oracle self-check (oracle/check_vs_reference.py) can
import the reference's own Python modules inside the build container, where numba,
torchvision, skimage, piq and tensorboard are not installed.
It never travels usefully to the GPU box: /root/reference does not exist there and
nothing under tests -m gpu / smoke() / bench.py imports this file.

Nothing is copied from the reference: the stubs below only stand in for *third-party*
packages the image lacks so that `import Model.model` / `import Recon.FBP_kernel` succeed.
"""
import math
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("IPDM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "Model", "model.py"))


def _lambda_ratio_numpy(I, idx, B, H, W, timesteps, lambda_):
    """numpy restatement of Model/model.py:340-351 (condition_lambda_ratio_cuda body).

    The reference kernel is numba-CUDA and cannot run on CPU; the arithmetic is
    float64 (python floats / math.cos) with a float32 exponent array and a float32 store."""
    s = 0.008
    lam = lambda_.astype(np.float64)
    a = []
    for x in (idx[0], idx[1], idx[2]):
        base = math.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
        a.append(np.power(base, lam))
    a1 = a[1] / a[0]
    a2 = a[2] / a[0]
    I[...] = (1 - (a2 / a1)).astype(I.dtype)


class _LambdaKernel:
    """Stands in for the @cuda.jit object: kernel[grid, block](args...) -> the numpy restatement above (fast; used by
    oracle/check_vs_reference.py at full size, where a per-pixel python loop takes minutes)."""

    def __getitem__(self, cfg):
        return _lambda_ratio_numpy


class _PyFuncLaunch:
    """Runs the BODY of the reference's own numba-CUDA kernel (its `.py_func`, Model/model.py:328-351) as plain
    python, one simulated thread at a time, with `cuda.grid` / `cuda.gridsize` of the stub module answering for that
    thread.  The kernel is a grid-stride loop, so only the threads that own at least one element are visited (all the
    others fall straight through their empty ranges).

    One adjustment, stated: numba types `float64 ** float32` as float64, whereas numpy>=2 *scalars* follow NEP 50 and
    would compute `python_float ** np.float32` in float32.  The exponent map is therefore handed to the body widened
    to float64 -- exactly the values numba's promotion produces -- and the result is stored into the caller's float32
    array as the kernel does."""

    def __init__(self, kernel, cuda_mod):
        self.body, self.cuda = kernel.py_func, cuda_mod

    def __getitem__(self, cfg):
        grid, block = cfg
        total = tuple(int(g) * int(b) for g, b in zip(grid, block))

        def launch(I, idx, B, H, W, timesteps, lambda_):
            lam64 = np.asarray(lambda_, dtype=np.float64)
            try:
                self.cuda.gridsize = lambda n: total
                for it in range(min(total[2], B)):
                    for iy in range(min(total[1], H)):
                        for ix in range(min(total[0], W)):
                            self.cuda.grid = lambda n, _t=(ix, iy, it): _t
                            self.body(I, idx, B, H, W, timesteps, lam64)
            finally:
                for name in ("grid", "gridsize"):
                    if hasattr(self.cuda, name):
                        delattr(self.cuda, name)
        return launch


def install():
    """Install stub modules and put the reference on sys.path. Idempotent."""
    if "numba" not in sys.modules or not hasattr(sys.modules["numba"], "_ipdm_stub"):
        nb = types.ModuleType("numba")
        nb._ipdm_stub = True
        nb.config = types.SimpleNamespace(NUMBA_DEFAULT_NUM_THREADS=8)

        def _jit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f

        nb.jit = _jit
        nb.prange = range
        cu = types.ModuleType("numba.cuda")

        class _NoCuda:
            def __init__(self, f):
                self.py_func = f

            def __getitem__(self, cfg):
                raise RuntimeError("numba-CUDA kernel unavailable on the CPU oracle")

        cu.jit = lambda f: _NoCuda(f)
        nb.cuda = cu
        sys.modules["numba"] = nb
        sys.modules["numba.cuda"] = cu
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tvt.ToTensor = object
        tv.transforms = tvt
        tvu = types.ModuleType("torchvision.utils")
        tvu.save_image = None
        tv.utils = tvu
        tv.__path__ = []
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
        sys.modules["torchvision.utils"] = tvu
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


_REF_LAMBDA_KERNEL = None


def load(lambda_kernel="py_func"):
    """Returns (Model.model, Recon.FBP_kernel) of the reference.  The numba-CUDA guidance kernel cannot be launched
    here; `lambda_kernel` chooses what answers `condition_lambda_ratio_cuda[grid, block](...)`:
      "py_func" (default, what the golden vectors are generated with): the reference's own kernel body executed per
                simulated thread (_PyFuncLaunch);
      "numpy":  the vectorised restatement (_lambda_ratio_numpy), for full-size runs."""
    global _REF_LAMBDA_KERNEL
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install()
    import Model.model as M
    import Recon.FBP_kernel as F
    if _REF_LAMBDA_KERNEL is None:
        _REF_LAMBDA_KERNEL = M.condition_lambda_ratio_cuda          # the stub's _NoCuda wrapper around the real body
    if lambda_kernel == "py_func":
        M.condition_lambda_ratio_cuda = _PyFuncLaunch(_REF_LAMBDA_KERNEL, sys.modules["numba.cuda"])
    else:
        M.condition_lambda_ratio_cuda = _LambdaKernel()
    return M, F


def _absent(what):
    def raiser(*a, **k):
        raise RuntimeError("%s is not available in this container" % what)
    return raiser


def load_curves():
    """curve_init / proj_curv_init / tensor_sharpen live in Utils/train_test_utils.py whose import
    needs more stubs (Windows .pyd, skimage, piq, tensorboard)."""
    install()
    for name, attrs in {
        "Recon.TASART2DNSL0": dict(recons_torch=_absent("Recon.TASART2DNSL0.recons_torch (Windows .pyd)"),
                                   proj_torch=_absent("Recon.TASART2DNSL0.proj_torch (Windows .pyd)")),
        "skimage": {},
        "skimage.metrics": dict(structural_similarity=None, peak_signal_noise_ratio=None),
        "piq": dict(vif_p=None, fsim=None),
        "torch.utils.tensorboard": dict(SummaryWriter=object),
    }.items():
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    import Utils.train_test_utils as U
    return U
