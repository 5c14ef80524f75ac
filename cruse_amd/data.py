"""Synthetic 16 kHz noisy/clean pairs: the reference's SynDataset is unfinished
(dataset/dataset.py ends mid-function) and the metric is quoted on synthetic data."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset


def synth_batch(batch: int, length: int, device, seed: int):
    """clean = 4 * one-pole low-pass (a = 0.95, 64-tap FIR) of 0.05*N(0,1) (speech-like tilt), noise = 0.1*N(0,1),
    noisy = clean + noise.  On a HIP device the filter and the mix are libcruse_hip kernels (cruse_onepole_fir,
    cruse_axpby); torch only draws the Gaussian samples.  On the CPU (DataLoader workers of SyntheticPairs) the same
    FIR runs through torch."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    white = 0.05 * torch.randn(batch, length, device=dev, generator=g)
    noise = 0.1 * torch.randn(batch, length, device=dev, generator=g)
    a, taps = 0.95, 64
    if dev.type == "cuda":
        from . import ops
        from ._lib import check, lib
        clean = torch.empty_like(white)
        check(lib.cruse_onepole_fir(ops._p(white), batch, length, a, taps, 4.0, ops._p(clean), ops._stream()))
        noisy = ops.axpby(torch.empty_like(clean), clean, noise, 1.0, 1.0)
        return noisy, clean
    w = (1 - a) * a ** torch.arange(taps - 1, -1, -1, dtype=torch.float32)
    clean = torch.nn.functional.conv1d(white.unsqueeze(1), w.view(1, 1, -1), padding=taps - 1)[..., :length].squeeze(1) * 4.0
    return (clean + noise).contiguous(), clean.contiguous()


def fir_causal(x: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """scipy.signal.fftconvolve(x, h)[:len(x)] per clip on the GPU (cruse_fir_causal).  x [B,L]; h [R] (shared) or [B,R]."""
    from . import ops
    from ._lib import check, lib
    x = x.contiguous().float()
    h = h.to(x.device).contiguous().float()
    B, L = x.shape
    if h.dim() == 1:
        R, hs = h.shape[0], 0
    elif h.dim() == 2 and h.shape[0] == B:
        R, hs = h.shape[1], h.shape[1]
    else:
        raise RuntimeError(f"fir_causal: taps {tuple(h.shape)} must be [R] or [{B}, R]")
    y = torch.empty_like(x)
    check(lib.cruse_fir_causal(ops._p(x), ops._p(h), hs, B, L, R, ops._p(y), ops._stream()))
    return y


def snr_mix(clean_y: torch.Tensor, noise_y: torch.Tensor, snr, target_dB_FS=None, target_dB_FS_floating_val=None, rir=None,
            rir_noise=None, eps: float = 1e-7, return_parts: bool = False):
    """SynDataset.snr_mix (dataset/dataset.py:235-264) with the reference's argument order, for a whole batch ON THE GPU:
    optional room impulse responses `rir` / `rir_noise` ([R] shared or [B,R]: fftconvolve(.)[:L], :245-248, as a direct FIR
    kernel), peak-normalise clean and noise, scale the noise by clean_rms / 10^(snr/20) / (noise_rms + eps), mix.
    clean_y, noise_y [B,L] (or [L]); snr: dB, scalar or [B].  Returns noisy (and the normalised clean / scaled noise with
    return_parts).  The reference's function ENDS after drawing `noisy_target_dB_FS` from
    np.random.randint(target_dB_FS - floating, target_dB_FS + floating) (the file is truncated there, :261-264): when both
    are given the draw is made -- once per clip of the batch -- to consume numpy's RNG exactly as B calls of the reference's
    per-clip function do, and nothing else is done with it."""
    from . import ops
    from ._lib import check, lib
    one = clean_y.dim() == 1
    c = clean_y.reshape(1, -1) if one else clean_y
    n = noise_y.reshape(1, -1) if one else noise_y
    if c.shape != n.shape or c.dim() != 2:
        raise RuntimeError(f"snr_mix: clean {tuple(clean_y.shape)} and noise {tuple(noise_y.shape)} must match ([B,L] or [L])")
    c = c.contiguous().float(); n = n.contiguous().float()
    if rir is not None:
        c = fir_causal(c, torch.as_tensor(rir))
    if rir_noise is not None:
        n = fir_causal(n, torch.as_tensor(rir_noise))
    B, L = c.shape
    snr_t = torch.as_tensor(snr, dtype=torch.float32, device=c.device).reshape(-1)
    if snr_t.numel() == 1:
        snr_t = snr_t.expand(B)
    snr_t = snr_t.contiguous()
    scratch = torch.empty(3 * B, device=c.device, dtype=torch.float64)
    noisy = torch.empty_like(c)
    co = torch.empty_like(c) if return_parts else None
    no = torch.empty_like(c) if return_parts else None
    check(lib.cruse_snr_mix(ops._p(c), ops._p(n), ops._p(snr_t), B, L, eps, ops._p(scratch), ops._p(co), ops._p(no), ops._p(noisy),
                            ops._stream()))
    if target_dB_FS is not None and target_dB_FS_floating_val is not None:
        import numpy as np
        # one draw PER CLIP: the reference's snr_mix is a per-clip function, so a batch of B clips consumes B draws there
        # (a single draw per batched call let identically seeded host RNG streams diverge for B > 1)
        np.random.randint(target_dB_FS - target_dB_FS_floating_val, target_dB_FS + target_dB_FS_floating_val,
                          size=None if one else B)                                                               # :261-264
    if one:
        noisy = noisy[0]; co = None if co is None else co[0]; no = None if no is None else no[0]
    return (noisy, co, no) if return_parts else noisy


class SyntheticPairs(Dataset):
    """[train_dataset] plug-in: path = "cruse_amd.data.SyntheticPairs", args = {num, length, seed}."""

    def __init__(self, num: int = 64, length: int = 64000, seed: int = 0):
        self.num, self.length, self.seed = num, length, seed

    def __len__(self):
        return self.num

    def __getitem__(self, i):
        noisy, clean = synth_batch(1, self.length, "cpu", self.seed * 100003 + i)
        return noisy[0], clean[0]
