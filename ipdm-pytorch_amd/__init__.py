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
