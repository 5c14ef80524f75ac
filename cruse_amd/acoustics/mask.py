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
    return _run(2, mask, K=float(K), C=float(C))


def decompress_cIRM(mask, K=10, limit=9.9):
    """mask.py:55-58."""
    return _run(3, mask, K=float(K), limit=float(limit))


def complex_mul(noisy_r, noisy_i, mask_r, mask_i):
    """mask.py:61-64."""
    return _run(4, noisy_r, noisy_i, mask_r, mask_i, two=True)
