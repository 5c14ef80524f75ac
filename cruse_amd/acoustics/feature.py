"""MI355X-native `train_base.acoustics.feature`: stft / istft with the reference signatures.

train_base/acoustics/feature.py:10-30 (stft) and :33-61 (istft); PreProcess.pre_stft layout
of utils/utils.py:389-412 as `pre_stft`.  Computation is the wavefront-shuffle FFT in
libcruse_hip.so (cruse_stft_fwd / cruse_istft_fwd); torch only packs the complex view.
"""
from __future__ import annotations

import torch

from .. import ops


def stft(y, n_fft, hop_length, win_length):
    """[B,L] -> complex64 [B,F,T] (feature.py:10-30)."""
    assert y.dim() == 2                                     # feature.py:21
    if win_length != n_fft:
        # the reference builds hann_window(n_fft) (:27) and torch.stft rejects a window != win_length
        raise RuntimeError(f"stft: window length n_fft={n_fft} must equal win_length={win_length}")
    re, im, _ = ops.stft(y.contiguous(), n_fft, hop_length)
    return torch.complex(re, im).transpose(1, 2)            # [B,T,F] -> [B,F,T] view


class _ISTFTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, re, im, n_fft, hop, length):
        ctx.meta = (re.shape[1], n_fft, hop)
        return ops.istft(re, im, n_fft, hop, length)

    @staticmethod
    def backward(ctx, dwave):
        T, n_fft, hop = ctx.meta
        dre, dim = ops.istft_bwd(dwave.contiguous(), T, n_fft, hop)
        return dre, dim, None, None, None


def istft_ri(re, im, n_fft, hop_length, length=None):
    """re, im [B,T,F] (frame-major) -> [B,L]; differentiable."""
    T = re.shape[1]
    if length is None:
        length = hop_length * (T - 1)
    return _ISTFTFn.apply(re.contiguous(), im.contiguous(), n_fft, hop_length, length)


def istft(features, n_fft, hop_length, win_length, length=None, use_mag_phase=False):
    """complex [B,F,T] (or (mag, phase)) -> [B,L] (feature.py:33-61)."""
    if win_length != n_fft:
        raise RuntimeError(f"istft: window length n_fft={n_fft} must equal win_length={win_length}")
    if use_mag_phase:                                        # feature.py:47-51
        assert isinstance(features, (tuple, list))
        mag, phase = features
        re = (mag * torch.cos(phase)).transpose(1, 2)
        im = (mag * torch.sin(phase)).transpose(1, 2)
    else:
        re = features.real.transpose(1, 2)
        im = features.imag.transpose(1, 2)
    return istft_ri(re, im, n_fft, hop_length, length)


def pre_stft(y, n_fft, hop_length, win_length, f_net=None):
    """PreProcess.pre_stft (utils/utils.py:389-412) on top of feature.stft's reflect padding:
    returns dict(real, imag [B,1,T,F], mag_net [B,1,T,f_net] = sqrt(re^2+im^2+1e-8)[..., :f_net])."""
    if win_length != n_fft:
        raise RuntimeError(f"pre_stft: window length n_fft={n_fft} must equal win_length={win_length}")
    F = n_fft // 2 + 1
    bins = F if f_net is None else f_net
    re, im, mag = ops.stft(y.contiguous(), n_fft, hop_length, mag_bins=bins, mag_eps=1e-8)
    return {"real": re.unsqueeze(1), "imag": im.unsqueeze(1), "mag_net": mag.unsqueeze(1)}
