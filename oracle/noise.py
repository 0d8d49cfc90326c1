"""CPU restatement of the device noise source (ipdm_randn, csrc/ddpm.hip): Philox4x32-10 keyed by (seed, global slice id,
draw index, element quad) + Box-Muller in float32.

The reference draws with torch.randn_like (Model/model.py:440,509), whose stream depends on the batch composition; the build
replaces it by a counter-based source so that a slice gets the same noise whatever the batch or the number of GPUs.  This file
restates THAT source for the tests (key layout and transform); replays of device runs still use the recorded draws, bit for bit.

TEST INFRASTRUCTURE ONLY -- imported by tests/; never by the product path."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten rounds on uint64 arrays holding 32-bit words (csrc/ddpm.hip: philox4x32_10)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    for r in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & _MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & _MASK
        c0, c1, c2, c3 = n0, p1 & _MASK, n2, p0 & _MASK
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def randn(seed, slice_id, draw, n):
    """The n float32 values of draw `draw` of global slice `slice_id` under `seed` (csrc/ddpm.hip: randn_kernel)."""
    nq = (n + 3) // 4
    q = np.arange(nq, dtype=np.uint64)
    sl, dr = np.uint64(slice_id), np.uint64(draw)
    c3 = ((sl >> np.uint64(32)) ^ ((q >> np.uint64(32)) << np.uint64(16)) ^ (dr >> np.uint64(32))) & _MASK
    c = philox4x32_10(q & _MASK, np.full(nq, dr & _MASK), np.full(nq, sl & _MASK), c3, int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    out = np.empty((nq, 4), np.float32)
    two_pi = np.float32(6.283185307179586)
    for h in range(2):
        u1 = ((c[2 * h] >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        u2 = ((c[2 * h + 1] >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        rad = np.sqrt(np.float32(-2.0) * np.log(u1))
        ang = two_pi * u2
        out[:, 2 * h] = rad * np.cos(ang)
        out[:, 2 * h + 1] = rad * np.sin(ang)
    return out.reshape(-1)[:n]
