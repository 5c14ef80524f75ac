"""MI355X-native `train_base.acoustics.mask` (mask.py:8-63): IRM / cIRM construction,
compress_cIRM / decompress_cIRM and complex_mul as HIP elementwise kernels (cruse_mask_ops)."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .._lib import check, lib

_p, _stream = ops._p, ops._stream
EPSILON = np.finfo(np.float32).eps          # train_base/constant.py


def _run(mode, a, b=None, c=None, d=None, K=10.0, C=0.1, limit=9.9, two=False, last=None):
    a = a.contiguous().float()
    ts = [None if t is None else t.contiguous().float() for t in (b, c, d)]
    shape = tuple(a.shape) + ((last,) if last else ())
    out = torch.empty(shape, device=a.device, dtype=torch.float32)
    out2 = torch.empty_like(out) if two else None
    check(lib.cruse_mask_ops(mode, _p(a), _p(ts[0]), _p(ts[1]), _p(ts[2]), a.numel(), K, C, limit, _p(out), _p(out2), _stream()))
    return (out, out2) if two else out


class _UnaryFn(torch.autograd.Function):
    """compress_cIRM / decompress_cIRM with their derivative kernels (cruse_mask_ops modes 7 / 8): the reference's versions
    are plain torch, and they are applied to network outputs during training."""

    @staticmethod
    def forward(ctx, x, fmode, bmode, K, C, limit):
        ctx.save_for_backward(x)
        ctx.cfg = (bmode, K, C, limit)
        return _run(fmode, x, K=K, C=C, limit=limit)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        bmode, K, C, limit = ctx.cfg
        return _run(bmode, x, c=g, K=K, C=C, limit=limit).view(x.shape), None, None, None, None, None


class _ComplexMulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nr, ni, mr, mi):
        ctx.save_for_backward(nr, ni, mr, mi)
        return _run(4, nr, ni, mr, mi, two=True)

    @staticmethod
    def backward(ctx, g1, g2):
        nr, ni, mr, mi = ctx.saved_tensors
        dnr = dni = dmr = dmi = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dnr, dni = _run(9, g1, g2, mr, mi, two=True)              # g * conj(mask)
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dmr, dmi = _run(9, g1, g2, nr, ni, two=True)              # g * conj(noisy)
        return dnr, dni, dmr, dmi


def build_ideal_ratio_mask(noisy_mag, clean_mag) -> torch.Tensor:
    """[B,F,T] x2 -> [B,F,T,1] = compress_cIRM(clean / (noisy + EPSILON)) (mask.py:8-21)."""
    return _run(0, noisy_mag, c=clean_mag, last=1)


def build_complex_ideal_ratio_mask(noisy: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
    """complex [B,F,T] x2 -> [B,F,T,2] (mask.py:24-40)."""
    return _run(1, noisy.real, noisy.imag, clean.real, clean.imag, last=2)


def compress_cIRM(mask, K=10, C=0.1):
    """mask.py:43-52 (tensor branch)."""
    if not torch.is_tensor(mask):
        raise RuntimeError("cruse_amd compress_cIRM: device tensors only (the numpy branch of the reference is host code)")
    return _UnaryFn.apply(mask, 2, 7, float(K), float(C), 9.9)


def decompress_cIRM(mask, K=10, limit=9.9):
    """mask.py:55-58."""
    return _UnaryFn.apply(mask, 3, 8, float(K), 0.1, float(limit))


def complex_mul(noisy_r, noisy_i, mask_r, mask_i):
    """mask.py:61-64."""
    return _ComplexMulFn.apply(noisy_r, noisy_i, mask_r, mask_i)
