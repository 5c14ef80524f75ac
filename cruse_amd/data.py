"""Synthetic 16 kHz noisy/clean pairs: the reference's SynDataset is unfinished
(dataset/dataset.py ends mid-function) and the metric is quoted on synthetic data."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset


def synth_batch(batch: int, length: int, device, seed: int):
    """clean = 0.05*N(0,1) through a one-pole low-pass (a=0.95, speech-like tilt), noise = 0.1*N(0,1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    white = 0.05 * torch.randn(batch, length, device=device, generator=g)
    # one-pole low-pass y[n] = a*y[n-1] + (1-a)*x[n], as a truncated FIR (64 taps) via cumulative products
    a = 0.95
    taps = (1 - a) * a ** torch.arange(63, -1, -1, device=device, dtype=torch.float32)
    clean = torch.nn.functional.conv1d(white.unsqueeze(1), taps.view(1, 1, -1), padding=63)[..., :length].squeeze(1)
    clean = clean * 4.0
    noise = 0.1 * torch.randn(batch, length, device=device, generator=g)
    return (clean + noise).contiguous(), clean.contiguous()


class SyntheticPairs(Dataset):
    """[train_dataset] plug-in: path = "cruse_amd.data.SyntheticPairs", args = {num, length, seed}."""

    def __init__(self, num: int = 64, length: int = 64000, seed: int = 0):
        self.num, self.length, self.seed = num, length, seed

    def __len__(self):
        return self.num

    def __getitem__(self, i):
        noisy, clean = synth_batch(1, self.length, "cpu", self.seed * 100003 + i)
        return noisy[0], clean[0]
