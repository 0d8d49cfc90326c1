"""Prints the end-to-end deviation of the smoke pipeline from the CPU oracle (max-abs, PSNR)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipdm_pytorch_amd
from ipdm_pytorch_amd.denoiser import smoke_pipeline
from oracle import pipeline as op, diffusion as od
got, inputs = smoke_pipeline("cuda:0")
want = op.smoke_pipeline_oracle(inputs)
d = np.abs(got - want)
print("legacy" if os.environ.get("IPDM_CONV_LEGACY") else "ws", "max|d| %.3e  mean|d| %.3e  rms %.3e  max|want| %.3f" % (d.max(), d.mean(), np.sqrt((d**2).mean()), np.abs(want).max()))
