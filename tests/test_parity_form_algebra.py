"""The identity behind the parity form of the Upsample layer (csrc/conv.hip conv_pack_weights_up2, conv_ws_kernel<2,1,...>,
conv_direct_up2_kernel), checked on the CPU in float64 independently of any kernel:

    conv3x3(nearest_2x(x), w, zero padding 1)[2y + a, 2x + b]
        = sum_{i, j in {0, 1}} W'[a][b][i][j] . x[y + i + a - 1, x + j + b - 1]        (x zero outside its grid)

with W'[a][b][i][j] the sum of the 3x3 taps (ky, kx) that land on that source pixel:
    a = 0:  i = 0 <- ky {0},     i = 1 <- ky {1, 2}          a = 1:  i = 0 <- ky {0, 1},   i = 1 <- ky {2}
(the same for b / kx).  This is the reference's Upsample (Model/model.py: F.interpolate(scale_factor=2, mode="nearest")
followed by nn.Conv2d(ch, ch, 3, padding=1)) with 4 instead of 9 multiply-adds per output."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROWS = {0: ({0}, {1, 2}), 1: ({0, 1}, {2})}          # parity -> taps folded into i = 0, i = 1


def pack_up2(w):
    """[Cout, Cin, 3, 3] -> [2, 2, Cout, Cin, 2, 2] (parity a, parity b, ..., i, j), float64 sums."""
    out = np.zeros((2, 2) + w.shape[:2] + (2, 2), dtype=np.float64)
    for a in (0, 1):
        for b in (0, 1):
            for i in (0, 1):
                for j in (0, 1):
                    for ky in ROWS[a][i]:
                        for kx in ROWS[b][j]:
                            out[a, b, :, :, i, j] += w[:, :, ky, kx]
    return out


@pytest.mark.parametrize("shape", [(1, 3, 5, 7, 4), (2, 2, 1, 1, 3), (1, 1, 4, 9, 1)])
def test_upsample_conv_equals_four_parity_convolutions(shape):
    B, C, H, W, Co = shape
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal((B, C, H, W))
    w = rng.standard_normal((Co, C, 3, 3))
    want = F.conv2d(F.interpolate(torch.from_numpy(x), scale_factor=2, mode="nearest"), torch.from_numpy(w), padding=1).numpy()
    wp = pack_up2(w)
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))              # source grid with the convolution's zero padding
    got = np.zeros_like(want)
    for a in (0, 1):
        for b in (0, 1):
            acc = np.zeros((B, Co, H, W))
            for i in (0, 1):
                for j in (0, 1):
                    win = xp[:, :, i + a:i + a + H, j + b:j + b + W]          # x[y + i + a - 1, x + j + b - 1]
                    acc += np.einsum("oc,bchw->bohw", wp[a, b, :, :, i, j], win)
            got[:, :, a::2, b::2] = acc
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    # 4 of the 9 products per output, and every 3x3 tap is used exactly once per parity
    for a in (0, 1):
        for b in (0, 1):
            assert np.allclose(wp[a, b].sum(axis=(-1, -2)), w.sum(axis=(-1, -2)))
