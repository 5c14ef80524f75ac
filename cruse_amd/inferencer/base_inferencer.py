"""Inference on the HIP path: noisy waveform -> magnitude -> unet_2 mask -> mask on the noisy spectrum (== enhanced
magnitude with the noisy phase) -> iSTFT, with the reference's RTF report and int16 scaling.

Mirrors train_base/inferencer/base_inferencer.py: `multi_channel_mag_to_mag` (:138-161: |STFT| -> model ->
enhanced_mag * (cos, sin)(noisy phase) -> istft(length=noisy.shape[-1])) and `__call__` (:163-200: batch size 1,
rtf = elapsed / (len / sr), 0.8 * int16 peak scaling).  CRUSE's unet_2 returns a sigmoid MASK on the first
`in_feat // 2 * 2` bins (model/cruse_net.py:164), so the enhanced magnitude is mask * |noisy| and
mag * cos(phase) == mask * real: the phase is never formed (PreProcess.masking, utils/utils.py:418-420).
"""
from __future__ import annotations

import time
from typing import Iterable, Optional, Tuple

import numpy as np
import torch

from ..acoustics import feature
from ..loss import enhanced_spectrum


class Inferencer:
    def __init__(self, model: torch.nn.Module, n_fft: int = 320, hop_length: int = 160, win_length: int = 320,
                 sr: int = 16000, device="cuda"):
        self.model = model.to(device).eval()
        self.n_fft, self.hop, self.win, self.sr = n_fft, hop_length, win_length, sr
        self.device = torch.device(device)
        self.f_net = (n_fft // 2 + 1) // 2 * 2            # unet_2 runs on in_feat // 2 * 2 bins

    @torch.no_grad()
    def mag_mask_to_wave(self, noisy: torch.Tensor) -> torch.Tensor:
        """noisy [B,L] (device) -> enhanced [B,L] (device)."""
        if noisy.dim() != 2:
            raise RuntimeError(f"Inferencer expects [B,L] waveforms, got {tuple(noisy.shape)}")
        spec = feature.pre_stft(noisy, self.n_fft, self.hop, self.win, f_net=self.f_net)
        mask = self.model(spec["mag_net"])                                   # [B,1,T,f_net]
        est = enhanced_spectrum(mask, spec["real"].squeeze(1), spec["imag"].squeeze(1))   # [B,T,F,2]
        return feature.istft_ri(est[..., 0], est[..., 1], self.n_fft, self.hop, length=noisy.shape[-1])

    @staticmethod
    def to_int16(enhanced: np.ndarray) -> np.ndarray:
        amp = np.iinfo(np.int16).max                                          # base_inferencer.py:183-185
        return np.int16(0.8 * amp * enhanced / np.max(np.abs(enhanced)))

    @torch.no_grad()
    def __call__(self, dataloader: Iterable[Tuple[torch.Tensor, list]], sink=None, log=print):
        """dataloader yields (noisy [1,L], [name]); returns [(name, rtf)], hands (name, int16 waveform) to sink."""
        out = []
        for noisy, name in dataloader:
            assert len(name) == 1, "The batch size of inference stage must 1."      # base_inferencer.py:173
            name = name[0]
            noisy = noisy.to(self.device)
            torch.cuda.synchronize()
            t1 = time.time()
            enhanced = self.mag_mask_to_wave(noisy).squeeze(0)
            torch.cuda.synchronize()
            t2 = time.time()
            enhanced = enhanced.cpu().numpy()
            if (np.abs(enhanced) > 1).any():
                log(f"Warning: enhanced is not in the range [-1, 1], {name}")
            wav = self.to_int16(enhanced)
            rtf = (t2 - t1) / (len(wav) * 1.0 / self.sr)                       # base_inferencer.py:187-190
            log(f"{name}, rtf: {rtf}")
            if sink is not None:
                sink(name, wav)
            out.append((name, rtf))
        return out
