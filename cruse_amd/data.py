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


class HostPoolPairs(Dataset):
    """[train_dataset] plug-in for a HOST dataset that costs the workers nothing per item: `pool` clips are synthesised once (per
    process) and item i is a view of clip i % pool -- the DataLoader's workers only collate.  With the trainer's pinned, double-buffered
    prefetcher (trainer_casual._Prefetcher) this measures what the reference's DataLoader path (tools/train_stand.py:46-57) can feed.
    path = "cruse_amd.data.HostPoolPairs", args = {num, length, seed, pool, dtype}."""

    def __init__(self, num: int = 2048, length: int = 64000, seed: int = 0, pool: int = 128, dtype: str = "float32"):
        self.num, self.length, self.seed, self.pool = num, length, seed, max(1, min(pool, num))
        # "float16": the samples are STORED as 16-bit values (as PCM audio is) -- half the bytes through the DataLoader's queue, the
        # staging copy and PCIe; the trainer widens them on the device (exactly)
        self.dtype = {"float32": torch.float32, "float16": torch.float16}[dtype]
        self._data = None

    def __len__(self):
        return self.num

    def _ensure(self):
        if self._data is None:
            noisy, clean = synth_batch(self.pool, self.length, "cpu", self.seed * 100003 + 17)
            self._data = (noisy.to(self.dtype).share_memory_(), clean.to(self.dtype).share_memory_())      # (forked workers read the parent's pages)
        return self._data

    def __getitem__(self, i):
        noisy, clean = self._ensure()
        return noisy[i % self.pool], clean[i % self.pool]


class DevicePairs(Dataset):
    """[train_dataset] plug-in for a DEVICE-RESIDENT dataset ("so real-data training is not host-bound"): clean and noise
    pools live in HBM and a batch is an index gather + the on-GPU SynDataset.snr_mix (dataset/dataset.py:235-264, cruse_snr_mix: peak
    normalisation, RMS-based SNR scaling, mix) at a per-clip SNR drawn from [snr_low, snr_high] dB -- no host tensor, no PCIe copy.
    The trainer recognises `device_resident` and asks for whole batches (device_batch) with the indices of the DataLoader's own
    sampler; the DataLoader object the reference flow builds around the dataset is never iterated.  __getitem__ still works (host
    copies of one pair) for anything that wants to look at an item.
    path = "cruse_amd.data.DevicePairs", args = {num, length, seed, pool, snr_low, snr_high}."""

    device_resident = True

    def __init__(self, num: int = 2048, length: int = 64000, seed: int = 0, pool: int = 128, snr_low: float = 0.0, snr_high: float = 20.0):
        self.num, self.length, self.seed, self.pool = num, length, seed, max(1, min(pool, num))
        self.snr_low, self.snr_high = float(snr_low), float(snr_high)
        self._pools = {}

    def __len__(self):
        return self.num

    def _ensure(self, device):
        device = torch.device(device)
        if device not in self._pools:
            g = torch.Generator(device=device).manual_seed(self.seed * 100003 + 29)
            _, clean = synth_batch(self.pool, self.length, device, self.seed * 100003 + 17)
            noise = torch.randn(self.pool, self.length, device=device, generator=g)
            snr = self.snr_low + (self.snr_high - self.snr_low) * torch.rand(self.num, device=device, generator=g)
            self._pools[device] = (clean, noise, snr)
        return self._pools[device]

    def device_batch(self, idx: torch.Tensor, device):
        """idx [B] int64 (host or device): -> (noisy, clean) [B, length] f32 on `device`, issued on the current stream"""
        clean_p, noise_p, snr = self._ensure(device)
        idx = idx.to(device, non_blocking=True)
        c = clean_p.index_select(0, idx % self.pool)
        n = noise_p.index_select(0, (idx * 7 + 3) % self.pool)
        noisy, clean, _ = snr_mix(c, n, snr.index_select(0, idx % self.num), return_parts=True)
        return noisy, clean

    def __getitem__(self, i):
        dev = torch.device("cuda", torch.cuda.current_device())
        noisy, clean = self.device_batch(torch.tensor([int(i)]), dev)
        return noisy[0].cpu(), clean[0].cpu()
