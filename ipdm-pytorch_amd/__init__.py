"""ipdm-pytorch_amd: MI355X-native (gfx950) implementation of IPDM's iterative partial-diffusion
sampling hot path, behind the reference's own call surface.

  progressive_domain_denoiser.update_opt()/progressive_denoiser()   (Utils/train_test_utils.py)
  GaussianDiffusion.guided_reverse_process                           (Model/model.py:517-642)
  UNetModel.forward                                                  (Model/model.py:283-310)
  FBP.convert                                                        (Recon/FBP_kernel.py:86-122)

All arithmetic runs in libipdm_hip.so (hand-written HIP, C ABI in include/ipdm_hip.h); torch is
used for device memory, streams and torch.distributed only.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from ._lib import IpdmError, lib  # noqa: F401

__all__ = ["IpdmError", "lib"]

_LAZY = {
    "progressive_domain_denoiser": "denoiser", "ResultTempDict": "denoiser", "miu2pixel": "denoiser",
    "default_cfg": "config", "cfg_load": "config",
    "GaussianDiffusion": "diffusion", "NoiseSource": "diffusion", "InjectedNoise": "diffusion",
    "UNetModel": "unet", "FBP": "fbp", "tensor_sharpen": "fbp",
}
__all__ += list(_LAZY)


def __getattr__(name):
    """The reference's class names at package level (torch is imported only when one is asked for)."""
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
