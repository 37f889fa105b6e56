"""oracle/fakenet.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A deterministic stand-in for the policy/value tower used ONLY to pin the
search arithmetic bit-for-bit: its outputs are an exact integer function of the
encoded planes, so the reference's own ``mctree.py`` (fed through a fake agent
on the CPU) and the HIP search (fed from the device encoder on the GPU) see
bit-identical float32 policies and values.  The real tower can never give that
(TensorFlow vs PyTorch, fp32 vs fp16).

All arithmetic is int64 without overflow (products < 2**62), so torch-CPU,
torch-ROCm and numpy agree exactly.  Policy entries are u24 * 2**-shift and the
value is s24 * 2**-23 - 1: both exactly representable in float32.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import numpy as np
import torch

_P = 2147483629           # prime < 2**31
_NPOL = 1968


def _splitmix(seed, n):
    """n pseudo-random 31-bit ints from splitmix64 (pure python, reproducible)."""
    out = np.zeros(n, dtype=np.int64)
    x = seed & 0xFFFFFFFFFFFFFFFF
    for i in range(n):
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        out[i] = z % (_P - 1) + 1
    return out


class FakeNet(object):
    """planes (B, 8, 8, C>=127) of 0/1 -> (policy f32 (B,1968), value f32 (B,)).

    ``prior_shift`` scales the priors by a power of two (24 -> priors in [0,1),
    wide trees; 31 -> priors in [0,1/128), deep trees) so both PUCT regimes get
    exercised.  ``quant`` coarsens the outputs to create exact ties.
    """

    # 10*f32(K) == 10*f32(K+1) when the product is rounded to float32 (numpy >= 2), but not when
    # it is a float64 product (numpy 1.x): priors K*2^-s and (K+1)*2^-s tie in one mode only.
    TIE_K = 13421774

    def __init__(self, seed=1, prior_shift=29, quant=0, device="cpu", tie=False):
        self.seed, self.prior_shift, self.quant, self.tie = seed, prior_shift, quant, tie
        self.device = torch.device(device)
        w = _splitmix(seed, 64 * 127)
        self.w = torch.from_numpy(w.reshape(8, 8, 127)).to(self.device)
        ab = _splitmix(seed ^ 0xABCDEF, 2 * (_NPOL + 1))
        self.a = torch.from_numpy(ab[:_NPOL + 1].copy()).to(self.device)
        self.b = torch.from_numpy(ab[_NPOL + 1:].copy()).to(self.device)

    def to(self, device):
        return FakeNet(self.seed, self.prior_shift, self.quant, device, self.tie)

    @torch.no_grad()
    def __call__(self, planes):
        x = planes[..., :127].to(torch.int64)
        h = (x * self.w).sum(dim=(1, 2, 3)) % _P            # < 8128 * 2**31 < 2**44
        mixed = (h[:, None] * self.a[None, :] + self.b[None, :]) % _P   # < 2**62
        mixed = (mixed * 48271) % _P                          # < 2**47
        u24 = mixed & 0xFFFFFF
        if self.quant:
            u24 = (u24 >> self.quant) << self.quant
        if self.tie:
            # two adjacent priors everywhere and one constant value (-0.5): exactly equal Q terms,
            # so the order of PUCT scores depends on how 10*prior is rounded
            u24 = torch.cat([self.TIE_K + (u24[:, :_NPOL] & 1),
                             torch.full_like(u24[:, _NPOL:], 1 << 22)], dim=1)
        pol = u24[:, :_NPOL].to(torch.float32) * (2.0 ** -self.prior_shift)
        val = u24[:, _NPOL].to(torch.float32) * (2.0 ** -23) - 1.0
        return pol, val
