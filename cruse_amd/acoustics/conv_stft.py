"""MI355X-native `train_base.acoustics.conv_stft.STFT` (conv_stft.py:8-129): Hamming-windowed
DFT-basis STFT with zero padding of win - hop on both sides, 161 bins, [B,T,F] outputs, and its inverse.

Repairs toward evident intent (the reference's constructor fails): `scipy.hamming` (:20) -> the symmetric Hamming window
it named; `nn.parameter` (:23) -> nn.Parameter.  istft (:100-129) as shipped reads `x[:, 0]` for both parts, concatenates
`spec_r` into the imaginary extension and adds the imaginary term with the wrong sign; the inverse built here is the one
those lines evidently meant -- the conjugate-symmetric inverse DFT, overlap-added and divided by the window sum -- and is
pinned by the round trip istft(stft(x)) == x.  The transform runs as a framed DFT (cruse_stft_framed), not a dense conv.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .feature import istft_framed, mag_phase, stft_framed


class STFT(nn.Module):
    def __init__(self, win_size=320, hop_size=160, requires_grad=False) -> None:
        super().__init__()
        if requires_grad:
            raise RuntimeError("cruse_amd conv_stft.STFT: the window is structural here, requires_grad must be False")
        self.win_size = win_size
        self.hop_size = hop_size
        self.n_overlap = self.win_size // self.hop_size
        self.requires_grad = requires_grad
        win = torch.from_numpy(np.hamming(self.win_size).astype(np.float32))          # == scipy.signal.hamming(N) (symmetric)
        win = F.relu(win)
        self.register_parameter("win", nn.Parameter(data=win, requires_grad=False))
        fourier_basis = np.fft.fft(np.eye(self.win_size))
        self.register_buffer("fourier_basis_r", torch.from_numpy(np.real(fourier_basis).astype(np.float32)))
        self.register_buffer("fourier_basis_i", torch.from_numpy(np.imag(fourier_basis).astype(np.float32)))
        self.register_buffer("idx", torch.tensor(range(self.win_size // 2 - 1, 0, -1), dtype=torch.long))
        self.eps = torch.finfo(torch.float32).eps

    def window(self, n_frames):
        """conv_stft.py:58-68: the hop-periodic sum of the window segments, tiled over the output length."""
        assert n_frames >= 2
        seg = sum([self.win[i * self.hop_size:(i + 1) * self.hop_size] for i in range(self.n_overlap)])
        seg = seg.unsqueeze(dim=-1).expand((self.hop_size, n_frames - self.n_overlap + 1))
        return seg.t().contiguous().view(-1).contiguous()

    def stft(self, sig):
        """sig [B,L] -> spec_r, spec_i, mag, pha [B,T,F]; T = (L + 2*(win-hop) - win)//hop + 1 (conv_stft.py:70-98)."""
        if sig.dim() != 2:
            raise RuntimeError(f"conv_stft.STFT.stft expects [B,L], got {tuple(sig.shape)}")
        pad = self.win_size - self.hop_size
        re, im = stft_framed(sig, self.win.data, self.win_size, self.hop_size, win_off=0, pad=pad, pad_mode="constant")
        mag, pha = mag_phase(re.detach(), im.detach())
        if re.requires_grad:
            mag = torch.sqrt(re ** 2 + im ** 2)                                       # keeps autograd on the magnitude
        return re, im, mag, pha

    def istft(self, x):
        """x [B,2,T,F] (real, imag planes) -> [B, hop*(T - n_overlap + 1)]."""
        spec_r, spec_i = x[:, 0].contiguous(), x[:, 1].contiguous()
        n_frames = spec_r.shape[1]
        L = self.hop_size * (n_frames - self.n_overlap + 1)
        ones = torch.ones(self.win_size, device=x.device)
        post = 1.0 / (self.window(n_frames).to(x.device) + self.eps)
        return istft_framed(spec_r, spec_i, ones, self.win_size, self.hop_size, win_off=0,
                            pad=self.win_size - self.hop_size, length=L, scale=1.0 / self.win_size, hermitian=True,
                            post_full=post)
