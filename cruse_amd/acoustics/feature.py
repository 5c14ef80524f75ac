"""MI355X-native `train_base.acoustics.feature`: stft / istft with the reference signatures.

train_base/acoustics/feature.py:10-30 (stft) and :33-61 (istft); PreProcess.pre_stft layout
of utils/utils.py:389-412 as `pre_stft`.  Computation is the wavefront-shuffle FFT in
libcruse_hip.so (cruse_stft_fwd / cruse_istft_fwd); torch only packs the complex view.
"""
from __future__ import annotations

import torch

import math

import torch.nn as nn

from .. import ops
from .._lib import check, lib

_p, _stream = ops._p, ops._stream


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


# ======================================================================================================================
# framed DFT with an arbitrary window (CustomSTFT / conv_stft / mtfaa STFT): cruse_stft_framed / cruse_istft_framed
# ======================================================================================================================
_PAD_MODE = {"constant": 0, "zeros": 0, "reflect": 1}


def _stft_framed_raw(wave, window, n_fft, hop, win_off, pad, pad_mode, T, scale):
    B, L = wave.shape
    F = n_fft // 2 + 1
    re = torch.empty(B, T, F, device=wave.device, dtype=torch.float32)
    im = torch.empty_like(re)
    check(lib.cruse_stft_framed(_p(wave), _p(window), B, L, n_fft, window.numel(), win_off, hop, pad, pad_mode, T, scale,
                                _p(re), _p(im), _stream()))
    return re, im


def _istft_framed_raw(re, im, window, post, n_fft, hop, win_off, pad, L, scale, hermitian):
    B, T, F = re.shape
    out = torch.empty(B, L, device=re.device, dtype=torch.float32)
    check(lib.cruse_istft_framed(_p(re), _p(im), _p(window), _p(post), B, T, n_fft, window.numel(), win_off, hop, pad, L, scale,
                                 1 if hermitian else 0, _p(out), _stream()))
    return out


class _StftFramedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wave, window, cfg):
        n_fft, hop, win_off, pad, pad_mode, T, scale = cfg
        ctx.cfg, ctx.L = cfg, wave.shape[1]
        ctx.save_for_backward(window)
        return _stft_framed_raw(wave.contiguous(), window, n_fft, hop, win_off, pad, pad_mode, T, scale)

    @staticmethod
    def backward(ctx, dre, dim):
        n_fft, hop, win_off, pad, pad_mode, T, scale = ctx.cfg
        if pad_mode != 0 and pad > 0:
            raise RuntimeError("stft_framed: gradients through reflect padding are not implemented")
        (window,) = ctx.saved_tensors
        dwave = _istft_framed_raw(dre.contiguous(), dim.contiguous(), window, None, n_fft, hop, win_off, pad, ctx.L, scale, False)
        return dwave, None, None


class _IstftFramedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, re, im, window, cfg):
        n_fft, hop, win_off, pad, L, scale = cfg
        ctx.cfg, ctx.T = cfg, re.shape[1]
        ctx.save_for_backward(window)
        return _istft_framed_raw(re.contiguous(), im.contiguous(), window, None, n_fft, hop, win_off, pad, L, scale, False)

    @staticmethod
    def backward(ctx, dwave):
        n_fft, hop, win_off, pad, L, scale = ctx.cfg
        (window,) = ctx.saved_tensors
        dre, dim = _stft_framed_raw(dwave.contiguous(), window, n_fft, hop, win_off, pad, 0, ctx.T, scale)
        return dre, dim, None, None


def stft_framed(wave, window, n_fft, hop, win_off=0, pad=0, pad_mode="constant", frames=None, scale=1.0):
    """wave [B,L] -> (re, im) [B,T,n_fft/2+1]: frame t = window * x_pad[t*hop + win_off - pad : ...] (see cruse_stft_framed).
    Differentiable wrt the wave (zero padding)."""
    if wave.dim() != 2:
        raise RuntimeError(f"stft_framed expects [B,L], got {tuple(wave.shape)}")
    L = wave.shape[1]
    T = frames if frames is not None else (L + 2 * pad - n_fft) // hop + 1
    if T <= 0:
        raise RuntimeError(f"stft_framed: {L} samples are shorter than one frame")
    return _StftFramedFn.apply(wave, window.contiguous().float(), (n_fft, hop, win_off, pad, _PAD_MODE[pad_mode], T, float(scale)))


def istft_framed(re, im, window, n_fft, hop, win_off=0, pad=0, length=None, scale=1.0, hermitian=False, post_full=None):
    """(re, im) [B,T,F] -> [B,L] overlap-add (see cruse_istft_framed).  Differentiable in the plain-adjoint form."""
    T = re.shape[1]
    L = length if length is not None else (T - 1) * hop + n_fft - 2 * pad
    if not hermitian and post_full is None:
        return _IstftFramedFn.apply(re, im, window.contiguous().float(), (n_fft, hop, win_off, pad, L, float(scale)))
    if re.requires_grad or im.requires_grad:
        raise RuntimeError("istft_framed: gradients are implemented for the plain adjoint form only")
    return _istft_framed_raw(re.contiguous(), im.contiguous(), window.contiguous().float(),
                             None if post_full is None else post_full.contiguous().float(), n_fft, hop, win_off, pad, L,
                             float(scale), hermitian)


class _MagPhaseFn(torch.autograd.Function):
    """(re, im) -> (sqrt(re^2 + im^2 + eps), atan2(im, re)) with both gradients (cruse_polar modes 0 / 2)."""

    @staticmethod
    def forward(ctx, re, im, eps):
        re = re.contiguous(); im = im.contiguous()
        m = torch.empty_like(re); p = torch.empty_like(re)
        check(lib.cruse_polar(0, _p(re), _p(im), None, None, re.numel(), eps, 1.0, _p(m), _p(p), _stream()))
        ctx.save_for_backward(re, im)
        ctx.eps = eps
        return m, p

    @staticmethod
    def backward(ctx, gm, gp):
        re, im = ctx.saved_tensors
        gm = None if gm is None else gm.contiguous()
        gp = None if gp is None else gp.contiguous()
        dr = torch.empty_like(re); di = torch.empty_like(re)
        check(lib.cruse_polar(2, _p(re), _p(im), _p(gm), _p(gp), re.numel(), ctx.eps, 1.0, _p(dr), _p(di), _stream()))
        return dr, di, None


class _PolarToRectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, p):
        m = m.contiguous(); p = p.contiguous()
        r = torch.empty_like(m); i = torch.empty_like(m)
        check(lib.cruse_polar(1, _p(m), _p(p), None, None, m.numel(), 0.0, 1.0, _p(r), _p(i), _stream()))
        ctx.save_for_backward(m, p)
        return r, i

    @staticmethod
    def backward(ctx, gr, gi):
        m, p = ctx.saved_tensors
        gr = torch.zeros_like(m) if gr is None else gr.contiguous()
        gi = torch.zeros_like(m) if gi is None else gi.contiguous()
        dm = torch.empty_like(m); dp = torch.empty_like(m)
        check(lib.cruse_polar(3, _p(m), _p(p), _p(gr), _p(gi), m.numel(), 0.0, 1.0, _p(dm), _p(dp), _stream()))
        return dm, dp


def mag_phase(re, im, eps: float = 0.0):
    """(re, im) -> ((re^2 + im^2 + eps) ** 0.5, atan2(im, re)) (feature.py:363-364); differentiable."""
    return _MagPhaseFn.apply(re, im, float(eps))


def polar_to_rect(m, p):
    """(mag, phase) -> (mag cos, mag sin) (feature.py:386-387); differentiable."""
    return _PolarToRectFn.apply(m, p)


def init_stft_kernel(frame_len, frame_hop, num_fft=None, window="sqrt_hann"):
    """feature.py:272-292 with torch.rfft (removed from torch) restated as the explicit real/imag DFT rows:
    K [fft_size + 2, 1, frame_len], rows 0..F-1 = cos(2 pi f n / N) w[n] / S_, rows F..2F-1 = -sin(.) w[n] / S_."""
    if window != "sqrt_hann":
        raise RuntimeError("Now only support sqrt hanning window in order to make signal perfectly reconstructed")
    fft_size = 2 ** math.ceil(math.log2(frame_len)) if not num_fft else num_fft
    w = torch.hann_window(frame_len) ** 0.5
    S_ = 0.5 * (fft_size * fft_size / frame_hop) ** 0.5
    n = torch.arange(frame_len, dtype=torch.float64)
    f = torch.arange(fft_size // 2 + 1, dtype=torch.float64)
    ang = 2 * math.pi * f[:, None] * n[None, :] / fft_size
    kernel = torch.cat([torch.cos(ang), -torch.sin(ang)], dim=0) / S_ * w.double()[None, :]
    return kernel.float().reshape(fft_size + 2, 1, frame_len)


class CustomSTFTBase(nn.Module):
    """feature.py:295-331.  K is kept as the (frozen) parameter the reference registers -- same state-dict key -- but the
    transform runs as a framed DFT on the HIP device with window = sqrt-Hann / S_ (K's structure), not as a dense conv."""

    def __init__(self, frame_len, frame_hop, window="sqrt_hann", num_fft=None):
        super().__init__()
        K = init_stft_kernel(frame_len, frame_hop, num_fft=num_fft, window=window)
        self.K = nn.Parameter(K, requires_grad=False)
        self.stride = frame_hop
        self.window = window
        self.frame_len = frame_len
        self.fft_size = K.shape[0] - 2
        self.scale = 1.0 / (0.5 * (self.fft_size * self.fft_size / frame_hop) ** 0.5)
        self.register_buffer("_win", torch.hann_window(frame_len) ** 0.5, persistent=False)

    def freeze(self):
        self.K.requires_grad = False

    def unfreeze(self):
        raise RuntimeError("cruse_amd CustomSTFT: the DFT kernel is structural here (framed DFT), it cannot be trained")

    def check_nan(self):
        pass

    def extra_repr(self):
        return "window={0}, stride={1}, requires_grad={2}, kernel_size={3[0]}x{3[2]}".format(
            self.window, self.stride, self.K.requires_grad, self.K.shape)


class CustomSTFT(CustomSTFTBase):
    """feature.py:334-366: x [N,S] or [N,1,S] -> m, p, r, i [N,F,T], T = (S - frame_len)//hop + 1 (no centre padding)."""

    def forward(self, x):
        if x.dim() not in [2, 3]:
            raise RuntimeError("Expect 2D/3D tensor, but got {:d}D".format(x.dim()))
        if x.dim() == 3:
            if x.shape[1] != 1:
                raise RuntimeError("CustomSTFT: expected N x 1 x S")
            x = x[:, 0]
        re, im = stft_framed(x, self._win, self.fft_size, self.stride, win_off=0, pad=0, scale=self.scale,
                             frames=(x.shape[-1] - self.frame_len) // self.stride + 1)
        r, i = re.transpose(1, 2), im.transpose(1, 2)                       # [N,F,T] views
        m, p = mag_phase(re, im)
        return m.transpose(1, 2), p.transpose(1, 2), r, i


class CustomISTFT(CustomSTFTBase):
    """feature.py:369-398: m, p [N,F,T] -> s [N,1,S], S = (T-1)*hop + frame_len (conv_transpose1d with the same K)."""

    def forward(self, m, p, squeeze=False):
        if p.dim() != m.dim() or p.dim() not in [2, 3]:
            raise RuntimeError("Expect 2D/3D tensor, but got {:d}D".format(p.dim()))
        if p.dim() == 2:
            p = torch.unsqueeze(p, 0)
            m = torch.unsqueeze(m, 0)
        r, i = polar_to_rect(m.transpose(1, 2), p.transpose(1, 2))           # frame-major [N,T,F]
        T = r.shape[1]
        s = istft_framed(r, i, self._win, self.fft_size, self.stride, win_off=0, pad=0,
                         length=(T - 1) * self.stride + self.frame_len, scale=self.scale).unsqueeze(1)
        if squeeze:
            s = torch.squeeze(s)
        return s
