"""MI355X-native blocks of `model/mtfaa.py` (BASELINE config 5): STFT.transform / inverse,
ComplexConv2d, ComplexLinearProjection, PhaseEncoder, TFCM_Block and TFCM on the [B,C,F,T] layout of that file.

Same constructor arguments, child modules and state-dict keys as model/mtfaa.py:8-210.  The file has no axial
attention, so this is everything config 5 can exercise: the STFT front end and the (complex /
depthwise-dilated) convolution stacks.  `Banks` needs the absent `spafe` package and is not part of the path.
Repairs: TFCM.__init__ calls `super(TFCM).__init__()` (:198) -> `super().__init__()`; STFT.inverse passes the window as
`win_length` (:35-36) -> (nfft, hop, win, window) with a complex input.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .._lib import check, lib
from ..nn_generic import HipConv2d, HipSequential, add, conv2d, to_f16, to_f32  # noqa: F401  (to_f16: where the fp16 part begins)

_p, _stream = ops._p, ops._stream


class STFT(nn.Module):
    """mtfaa.py:8-37: torch.stft(inp, nfft, hop, win, window) (center=True, reflect) -> [B,2,F,T]."""

    def __init__(self, win_len, hop_len, fft_len, win_type) -> None:
        super().__init__()
        self.win, self.hop = win_len, hop_len
        self.nfft = fft_len
        window = {"hann": torch.hann_window(win_len), "hamm": torch.hamming_window(win_len)}
        assert win_type in window.keys()
        self.window = window[win_type]
        self._cache = {}            # device copies of the window / inverse envelopes: made once, not per call (a host-to-device
                                    # copy per call also rules out HIP-graph capture of a step that contains the transform)

    def _window_on(self, device):
        key = ("w", str(device))
        if key not in self._cache:
            self._cache[key] = self.window.to(device)
        return self._cache[key]

    def transform(self, inp):
        from ..acoustics.feature import stft_framed
        w = self._window_on(inp.device)
        re, im = stft_framed(inp, w, self.nfft, self.hop, win_off=(self.nfft - self.win) // 2, pad=self.nfft // 2,
                             pad_mode="reflect", frames=1 + inp.shape[-1] // self.hop)
        return torch.stack([re.transpose(1, 2), im.transpose(1, 2)], dim=1)          # "b f t c -> b c f t"

    def inverse(self, real, imag):
        """real, imag: [B,F,T] -> [B, hop*(T-1)] (torch.istft, center=True)."""
        from ..acoustics.feature import istft_framed
        w = self._window_on(real.device)
        T = real.shape[-1]
        L = self.hop * (T - 1)
        off = (self.nfft - self.win) // 2
        key = ("env", T, str(real.device))
        if key not in self._cache:
            if len(self._cache) >= 8:                        # (variable clip lengths: keep a handful of envelopes, not one per length for ever)
                self._cache.pop(next(iter(self._cache)))
            # window-square overlap envelope: periodic in hop away from the clip edges; torch.istft divides by it
            # (one overlap-add of T copies of w^2 as a fold, once per clip length)
            w2 = torch.nn.functional.pad(self.window ** 2, (off, self.nfft - self.win - off))
            env = torch.nn.functional.fold(w2.view(1, self.nfft, 1).expand(1, self.nfft, T).contiguous(), (1, self.nfft + self.hop * (T - 1)),
                                           (1, self.nfft), stride=(1, self.hop)).view(-1)
            env = env[self.nfft // 2:self.nfft // 2 + L]
            self._cache[key] = (1.0 / env).to(real.device)
        return istft_framed(real.transpose(1, 2).contiguous(), imag.transpose(1, 2).contiguous(), w, self.nfft, self.hop,
                            win_off=off, pad=self.nfft // 2, length=L, scale=1.0 / self.nfft, hermitian=True,
                            post_full=self._cache[key])


class ComplexConv2d(nn.Module):
    """mtfaa.py:39-107.  The four real convolutions (real/imag weights on real/imag halves, :94-101) run as ONE real
    convolution with the block weight [[Wr, -Wi], [Wi, Wr]] and bias [br - bi, br + bi] (complex_axis = 1)."""

    def __init__(self, in_channels, out_channels, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), dilation=1, groups=1,
                 casual=True, complex_axis=1):
        super().__init__()
        self.in_channels = in_channels // 2
        self.out_channels = out_channels // 2
        self.kernel_size = kernel_size
        self.stride = stride
        self.padding = padding
        self.causal = casual
        self.groups = groups
        self.dilation = dilation
        self.complex_axis = complex_axis
        if complex_axis != 1 or groups != 1:
            raise RuntimeError("cruse_amd ComplexConv2d: complex_axis = 1 and groups = 1 (the forms model/mtfaa.py uses)")
        self.real_conv = nn.Conv2d(self.in_channels, self.out_channels, kernel_size, self.stride,
                                   padding=(self.padding[0], 0), dilation=self.dilation, groups=self.groups)
        self.imag_conv = nn.Conv2d(self.in_channels, self.out_channels, kernel_size, self.stride,
                                   padding=(self.padding[0], 0), dilation=self.dilation, groups=self.groups)
        nn.init.normal_(self.real_conv.weight.data, std=0.05)
        nn.init.normal_(self.imag_conv.weight.data, std=0.05)
        nn.init.normal_(self.real_conv.bias, 0.)
        nn.init.normal_(self.imag_conv.bias, 0.)

    def forward(self, inputs, pre_pad=(0, 0, 0, 0)):
        """pre_pad (top, bottom, left, right): a ConstantPad2d in front of this layer, folded into the gather."""
        wr, wi = self.real_conv.weight, self.imag_conv.weight
        w = torch.cat([torch.cat([wr, -wi], dim=1), torch.cat([wi, wr], dim=1)], dim=0)       # parameter plumbing
        br, bi = self.real_conv.bias, self.imag_conv.bias
        b = torch.cat([br - bi, br + bi], dim=0)
        pw = self.padding[1]
        pl, pr = (pw, 0) if (pw != 0 and self.causal) else (pw, pw)                          # :81-85
        ph = self.padding[0]
        pad = (pre_pad[0] + ph, pre_pad[1] + ph, pre_pad[2] + pl, pre_pad[3] + pr)
        d = self.dilation if isinstance(self.dilation, (tuple, list)) else (self.dilation, self.dilation)
        return conv2d(inputs, w, b, self.stride, d, pad, 1)


def complex_cat(inps, dim=1):
    reals, imags = [], []
    for inp in inps:
        real, imag = inp.chunk(2, dim)
        reals.append(real)
        imags.append(imag)
    return torch.cat(reals, dim), torch.cat(imags, dim)


class _AmpPowFn(torch.autograd.Function):
    """sqrt(real^2 + imag^2 + eps) ** alpha (mtfaa.py:136-137,162)."""

    @staticmethod
    def forward(ctx, real, imag, eps, alpha):
        real = real.contiguous(); imag = imag.contiguous()
        out = torch.empty_like(real)
        check(lib.cruse_polar(0, _p(real), _p(imag), None, None, real.numel(), eps, alpha, _p(out), None, _stream()))
        ctx.save_for_backward(real, imag)
        ctx.eps, ctx.alpha = eps, alpha
        return out

    @staticmethod
    def backward(ctx, g):
        real, imag = ctx.saved_tensors
        dr = torch.empty_like(real); di = torch.empty_like(real)
        check(lib.cruse_polar(2, _p(real), _p(imag), _p(g.contiguous()), None, real.numel(), ctx.eps, ctx.alpha, _p(dr), _p(di), _stream()))
        return dr, di, None, None


class ComplexLinearProjection(nn.Module):
    """mtfaa.py:122-138."""

    def __init__(self, cin):
        super().__init__()
        self.clp = ComplexConv2d(cin, cin)

    def forward(self, real, imag, alpha: float = 1.0):
        outputs = to_f32(self.clp(torch.cat((real, imag), 1)))       # (the magnitude / power kernel runs on f32 in either mode)
        real, imag = outputs.chunk(2, dim=1)
        return _AmpPowFn.apply(real, imag, 1e-8, alpha)


class _PadThenComplexConv(nn.Sequential):
    """nn.Sequential(ConstantPad2d((2,0,0,0)), ComplexConv2d) of mtfaa.py:147-149 (keys `0`, `1.*`), pad folded."""

    def forward(self, x):
        l, r, t, b = self[0].padding
        return self[1](x, pre_pad=(t, b, l, r))


class PhaseEncoder(nn.Module):
    """mtfaa.py:141-163: per-signal causal complex conv (1,3) along T -> complex cat -> complex 1x1 -> |.| ** alpha."""

    def __init__(self, cout, n_sig, cin=2, alpha=0.5):
        super().__init__()
        self.complexnn = nn.ModuleList()
        for _ in range(n_sig):
            self.complexnn.append(_PadThenComplexConv(nn.ConstantPad2d((2, 0, 0, 0), 0.0), ComplexConv2d(cin, cout, (1, 3))))
        self.clp = ComplexLinearProjection(cout * n_sig)
        self.alpha = alpha

    def forward(self, cspecs):
        outs = [layer(cspecs[idx]) for idx, layer in enumerate(self.complexnn)]
        real, imag = complex_cat(outs, dim=1)
        return self.clp(real, imag, alpha=self.alpha)             # amp ** alpha fused into the magnitude kernel


class TFCM_Block(nn.Module):
    """mtfaa.py:166-193: 1x1 conv + BN + PReLU -> depthwise (3,3) conv dilated along T, causal -> BN + PReLU -> 1x1, + input."""

    def __init__(self, cin=24, K=(3, 3), dila=1, causal=True):
        super().__init__()
        self.pconv1 = HipSequential(nn.Conv2d(cin, cin, kernel_size=(1, 1)), nn.BatchNorm2d(cin), nn.PReLU(cin))
        dila_pad = dila * (K[1] - 1)
        if causal:
            self.dila_conv = HipSequential(nn.ConstantPad2d((dila_pad, 0, 1, 1), 0.0),
                                           nn.Conv2d(cin, cin, K, 1, dilation=(1, dila), groups=cin),
                                           nn.BatchNorm2d(cin), nn.PReLU(cin))
        else:
            # the reference passes a third positional argument to ConstantPad2d (:181) and fails; evident intent: value 0
            self.dila_conv = HipSequential(nn.ConstantPad2d((dila_pad // 2, dila_pad // 2, 1, 1), 0.0),
                                           nn.Conv2d(cin, cin, K, 1, dilation=(1, dila)),
                                           nn.BatchNorm2d(cin), nn.PReLU(cin))
        self.pconv2 = HipConv2d(cin, cin, kernel_size=(1, 1))
        self.causal = causal
        self.dila_pad = dila_pad

    def forward(self, inps):
        # (the input also feeds the residual: it is handed through pconv1's autograd node, so that the residual path's gradient is added inside
        #  pconv1's data-gradient kernel instead of by an accumulation pass)
        outs, skip = self.pconv1.forward_tap(inps)
        outs = self.dila_conv(outs)
        return self.pconv2.forward_add(outs, skip)


class TFCM(nn.Module):
    """mtfaa.py:196-209 (repair: super().__init__())."""

    def __init__(self, cin=24, K=(3, 3), tfcm_layer=6, causal=True) -> None:
        super().__init__()
        self.tfcm = nn.ModuleList()
        for idx in range(tfcm_layer):
            self.tfcm.append(TFCM_Block(cin, K, 2 ** idx, causal=causal))

    def forward(self, inp):
        out = inp
        for idx in range(len(self.tfcm)):
            out = self.tfcm[idx](out)
        return out
