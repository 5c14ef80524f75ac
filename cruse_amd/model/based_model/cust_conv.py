"""MI355X-native `model.based_model.cust_conv` conv / grouped-GRU blocks (8f item 2).

Same constructor arguments, child modules (hence state-dict keys and default initialisation) and forward semantics as
model/based_model/cust_conv.py:15-184 (Conv2dNormAct, ConvTranspose2dNormAct, convkxf with its normal / transposed /
upsample modes and the depthwise + 1x1 form, FreqUpsample) and :250-416 (GroupedGRULayer, GroupGRU with the inter-layer
shuffle and add_outputs).  The children are parameter containers; forward runs the fused HIP kernels of
cruse_amd/nn_generic.py (conv blocks) and the persistent GRU recurrence kernels (cruse_amd/csrc/gru.hip).
Layout is the reference's [B,C,T,F] / [B,T,I].  No CPU path.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import List, Optional, Tuple

import torch
from torch import Tensor, nn

from ... import ops
from ...nn_generic import FreqUpsample, HipSequential  # noqa: F401  (FreqUpsample is part of this module's surface)


class Conv2dNormAct(HipSequential):
    """cust_conv.py:15-62.  [B,C,T,F]; causal pad of kernel_size[0]-1 frames on top, frequency stride `fstride`."""

    def __init__(self, in_ch, out_ch, kernel_size, fstride=1, dilation=1, fpad=True, bias=True, separable=False,
                 norm_layer=torch.nn.BatchNorm2d, activation_layer=torch.nn.ReLU):
        lookahead = 0
        kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        fpad_ = kernel_size[1] // 2 + dilation - 1 if fpad else 0                     # :33-36
        pad = (0, 0, kernel_size[0] - 1 - lookahead, lookahead)                        # :37
        layers = []
        if any(x > 0 for x in pad):
            layers.append(nn.ConstantPad2d(pad, 0.0))
        groups = math.gcd(in_ch, out_ch) if separable else 1                           # :41
        if groups == 1:
            separable = False
        if max(kernel_size) == 1:
            separable = False
        layers.append(nn.Conv2d(in_ch, out_ch, kernel_size, padding=(0, fpad_), stride=(1, fstride),
                                dilation=(1, dilation), groups=groups, bias=bias))    # :46-55
        if separable:
            layers.append(nn.Conv2d(out_ch, out_ch, kernel_size=1, bias=False))        # :56-57
        if norm_layer is not None:
            layers.append(norm_layer(out_ch))
        if activation_layer is not None:
            layers.append(activation_layer())
        super().__init__(*layers)


class ConvTranspose2dNormAct(HipSequential):
    """cust_conv.py:65-111."""

    def __init__(self, in_ch, out_ch, kernel_size, fstride=1, dilation=1, fpad=True, bias=True, separable=False,
                 norm_layer=torch.nn.BatchNorm2d, activation_layer=torch.nn.ReLU):
        lookahead = 0
        kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
        fpad_ = kernel_size[1] // 2 if fpad else 0                                     # :79-82
        pad = (0, 0, kernel_size[0] - 1 - lookahead, lookahead)
        layers = []
        if any(x > 0 for x in pad):
            layers.append(nn.ConstantPad2d(pad, 0.0))
        groups = math.gcd(in_ch, out_ch) if separable else 1
        if groups == 1:
            separable = False
        layers.append(nn.ConvTranspose2d(in_ch, out_ch, kernel_size=kernel_size,
                                         padding=(kernel_size[0] - 1, fpad_ + dilation - 1), output_padding=(0, fpad_),
                                         stride=(1, fstride), dilation=(1, dilation), groups=groups, bias=bias))  # :93-104
        if separable:
            layers.append(nn.Conv2d(out_ch, out_ch, kernel_size=1, bias=False))
        if norm_layer is not None:
            layers.append(norm_layer(out_ch))
        if activation_layer is not None:
            layers.append(activation_layer())
        super().__init__(*layers)


def convkxf(in_ch, out_ch, k=1, f=3, fstride=2, lookahead=0, batch_norm=False, act=None, mode="normal", depthwise=True,
            complex_in=False):
    """cust_conv.py:114-174; `act` defaults to nn.ReLU(inplace=True) as there."""
    if act is None:
        act = torch.nn.ReLU(inplace=True)
    bias = batch_norm is False
    assert f % 2 == 1
    stride = 1 if f == 1 else (1, fstride)
    if out_ch is None:
        out_ch = in_ch * 2 if mode == "normal" else in_ch // 2
    fpad = (f - 1) // 2
    convpad = (0, fpad)
    modules = []
    pad = [0, 0, k - 1 - lookahead, lookahead]
    if any(p > 0 for p in pad):
        modules.append(("pad", nn.ConstantPad2d(pad, 0.0)))
    groups = min(in_ch, out_ch) if depthwise else 1
    if in_ch % groups != 0 or out_ch % groups != 0:
        groups = 1
    if complex_in and groups % 2 == 0:
        groups //= 2
    convkwargs = {"in_channels": in_ch, "out_channels": out_ch, "kernel_size": (k, f), "stride": stride, "groups": groups,
                  "bias": bias}
    if mode == "normal":
        modules.append(("sconv", nn.Conv2d(padding=convpad, **convkwargs)))
    elif mode == "transposed":
        modules.append(("sconv", nn.ConvTranspose2d(padding=(k - 1, fpad), output_padding=convpad, **convkwargs)))
    elif mode == "upsample":
        modules.append(("upsample", FreqUpsample(fstride)))
        convkwargs["stride"] = 1
        modules.append(("sconv", nn.Conv2d(padding=convpad, **convkwargs)))
    else:
        raise NotImplementedError()
    if groups > 1:
        modules.append(("1x1conv", nn.Conv2d(out_ch, out_ch, 1, bias=False)))
    if batch_norm:
        modules.append(("norm", nn.BatchNorm2d(out_ch)))
    modules.append(("act", act))
    return HipSequential(OrderedDict(modules))


# ======================================================================================================================
# grouped GRUs
# ======================================================================================================================
class _GroupedGruFn(torch.autograd.Function):
    """g independent nn.GRU(I/g -> H/g) over feature slices, outputs concatenated (cust_conv.py:303-325), started from the
    state h0 [B, H] ("cat" layout) the caller passes -- DETACHED, as the reference does (:318-319): no gradient flows into
    it -- or from 0.  params = (w_ih_0, w_hh_0, b_ih_0, b_hh_0, w_ih_1, ...).  Exact-f32 MFMA kernels (v_mfma_f32_16x16x4_f32)."""

    @staticmethod
    def forward(ctx, x, g, h0, *params):
        x = x.contiguous()
        B, T, I = x.shape
        Ig = I // g
        Hg = params[1].shape[1]
        H = g * Hg
        rows = B * T
        gi = torch.empty(B, T, g * 3 * Hg, device=x.device, dtype=torch.float32)
        for i in range(g):
            w_ih, b_ih = params[4 * i], params[4 * i + 2]
            ops.gemm(False, True, rows, 3 * Hg, Ig, x, i * Ig, I, w_ih.contiguous(), 0, Ig, gi, i * 3 * Hg, 3 * H, bias=b_ih,
                     prec="f32")
        w_hh = [params[4 * i + 1].contiguous() for i in range(g)]
        b_hh = [params[4 * i + 3].contiguous() for i in range(g)]
        need = any(ctx.needs_input_grad)
        h, coef, an, z = ops.gru_seq_fwd(gi, w_hh, b_hh, B, T, g, Hg, "f32", save=need, h0=h0)
        ctx.g, ctx.dims = g, (B, T, I, Ig, Hg)
        ctx.h0 = h0
        ctx.save_for_backward(x, h, coef, an, z, *params)
        return h

    @staticmethod
    def backward(ctx, dout):
        x, h, coef, an, z, *params = ctx.saved_tensors
        g = ctx.g
        B, T, I, Ig, Hg = ctx.dims
        H, rows = g * Hg, B * T
        w_hh = [params[4 * i + 1].contiguous() for i in range(g)]
        dh = ops.gru_seq_bwd(dout.contiguous(), w_hh, coef, z, B, T, g, Hg, "f32")
        dgi, dgh = ops.gru_gate_grads(dh, coef, an, rows, g, Hg, "f32")
        grads = []
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        sk = max(1, min(8, rows // 512))
        for i in range(g):
            w_ih = params[4 * i].contiguous()
            dw_ih = torch.zeros_like(w_ih); dw_hh = torch.zeros_like(w_hh[i])
            db_ih = torch.zeros(3 * Hg, device=x.device); db_hh = torch.zeros(3 * Hg, device=x.device)
            ops.gemm(True, False, 3 * Hg, Ig, rows, dgi, i * 3 * Hg, 3 * H, x, i * Ig, I, dw_ih, 0, Ig, accumulate=True,
                     splitk=sk, prec="f32")
            ops.gemm(True, False, 3 * Hg, Hg, rows, dgh, i * 3 * Hg, 3 * H, h, i * Hg, H, dw_hh, 0, Hg, accumulate=True,
                     splitk=sk, b_shift_T=T, prec="f32")
            if ctx.h0 is not None:
                # frame 0's h_{t-1} is h0, not 0: dW_hh += dgh[:, 0]^T h0 (K = B; rows of frame 0 are T frames apart)
                ops.gemm(True, False, 3 * Hg, Hg, B, dgh, i * 3 * Hg, T * 3 * H, ctx.h0, i * Hg, H, dw_hh, 0, Hg, accumulate=True,
                         prec="f32")
            ops.col_sum(dgi, i * 3 * Hg, rows, 3 * Hg, 3 * H, db_ih)
            ops.col_sum(dgh, i * 3 * Hg, rows, 3 * Hg, 3 * H, db_hh)
            if dx is not None:
                ops.gemm(False, False, rows, Ig, 3 * Hg, dgi, i * 3 * Hg, 3 * H, w_ih, 0, Ig, dx, i * Ig, I, prec="f32")
            grads += [dw_ih, dw_hh, db_ih, db_hh]
        return (dx, None, None) + tuple(grads)



class GroupedGRULayer(nn.Module):
    """cust_conv.py:250-325."""

    def __init__(self, input_size: int, hidden_size: int, groups: int, batch_first: bool = True, bias=True,
                 dropout: float = 0, bidirectional=False):
        super().__init__()
        assert input_size % groups == 0
        assert hidden_size % groups == 0
        # dropout: nn.GRU applies it BETWEEN stacked layers only; every group's nn.GRU here has one layer (cust_conv.py:286-287), so the
        # reference's argument has no effect either (torch warns) -- it is accepted and kept in the sub-modules' kwargs
        kwargs = {"bias": bias, "batch_first": batch_first, "dropout": dropout, "bidirectional": bidirectional}
        self.input_size = input_size // groups
        self.hidden_size = hidden_size // groups
        self.out_size = hidden_size
        self.bidirectional = bidirectional
        self.num_directions = 2 if bidirectional else 1
        self.groups = groups
        self.batch_first = batch_first
        assert (self.hidden_size % groups) == 0, "Hidden size must be divisible by groups"       # :284-285
        if self.hidden_size % 32:
            raise RuntimeError(f"cruse_amd GroupedGRULayer: hidden size per group {self.hidden_size} must be a multiple of 32")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                 # (torch: "dropout expects num_layers greater than 1")
            self.layers = nn.ModuleList(nn.GRU(self.input_size, self.hidden_size, **kwargs) for _ in range(groups))

    def flatten_parameters(self):
        pass

    def get_h0(self, batch_size: int = 1, device: torch.device = torch.device("cpu")):
        return torch.zeros(self.groups * self.num_directions, batch_size, self.hidden_size, device=device)

    def _direction(self, x: Tensor, h0c: Optional[Tensor], suffix: str) -> Tensor:
        params = []
        for layer in self.layers:
            w_ih, w_hh = getattr(layer, "weight_ih_l0" + suffix), getattr(layer, "weight_hh_l0" + suffix)
            if layer.bias:
                b_ih, b_hh = getattr(layer, "bias_ih_l0" + suffix), getattr(layer, "bias_hh_l0" + suffix)
            else:
                # nn.GRU(bias=False) (cust_conv.py:262,272) has no bias parameters: the kernels take zero vectors (one shared
                # buffer, no gradient asked for -- what _GroupedGruFn returns for them is dropped by autograd)
                b_ih = b_hh = self._zero_bias(w_ih.device)
            params += [w_ih, w_hh, b_ih, b_hh]
        return _GroupedGruFn.apply(x, self.groups, h0c, *params)

    def _zero_bias(self, device) -> Tensor:
        z = getattr(self, "_zb", None)
        if z is None or z.device != device:
            z = torch.zeros(3 * self.hidden_size, device=device)
            object.__setattr__(self, "_zb", z)                # (not a buffer: the state dict stays the reference's)
        return z

    def forward(self, input: Tensor, h0: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        # state [G*D, B, H/g] (:305-306; group-major, direction-minor) -> the kernel's [B, G*H/g] "cat" layout per direction;
        # detached like the reference's (:319)
        if not self.batch_first:
            # [T, B, I] (cust_conv.py:306): the kernels are batch-major -- one device transposition in, one out; the state
            # layout [G*D, B, H/g] does not depend on batch_first (torch)
            y, hn = self._forward_bf(input.transpose(0, 1), h0)
            return y.transpose(0, 1).contiguous(), hn
        return self._forward_bf(input, h0)

    def _forward_bf(self, input: Tensor, h0: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
        B, D, g, Hg = input.shape[0], self.num_directions, self.groups, self.hidden_size
        h0d = [None] * D
        if h0 is not None:
            if tuple(h0.shape) != (g * D, B, Hg):
                raise RuntimeError(f"GroupedGRULayer: state {tuple(h0.shape)} != {(g * D, B, Hg)}")
            st = h0.detach().to(input.device, torch.float32).view(g, D, B, Hg)
            h0d = [st[:, d].transpose(0, 1).reshape(B, -1).contiguous() for d in range(D)]
        out = self._direction(input, h0d[0], "")
        if not self.bidirectional:
            # final states [G, B, H/g] = the last frame of every group's slice (cust_conv.py:323)
            return out, out[:, -1, :].reshape(B, g, Hg).transpose(0, 1).contiguous()
        # the reverse direction (nn.GRU(bidirectional=True): weights *_reverse) is the same recurrence on the time-reversed
        # sequence, its outputs reversed back; a group's output is [forward | reverse] (torch), the groups are concatenated
        # after that (:322), and a group's two final states are h_{T-1} of the forward and h_0 of the reverse direction
        rev = torch.flip(self._direction(torch.flip(input, dims=(1,)), h0d[1], "_reverse"), dims=(1,))
        T = out.shape[1]
        y = torch.stack((out.view(B, T, g, Hg), rev.view(B, T, g, Hg)), dim=3).reshape(B, T, g * 2 * Hg)
        hf = out[:, -1, :].reshape(B, g, Hg)
        hr = rev[:, 0, :].reshape(B, g, Hg)
        return y, torch.stack((hf, hr), dim=2).permute(1, 2, 0, 3).reshape(g * 2, B, Hg).contiguous()


class GroupGRU(nn.Module):
    """cust_conv.py:328-416: stacked GroupedGRULayers with the inter-layer feature shuffle
    view(B,T,-1,g).transpose(2,3) (:408-410) and optional add_outputs."""

    def __init__(self, input_size, hidden_size, num_layers=1, groups=4, bias=True, batch_first=True, dropout=0.,
                 bidirectional=False, shuffle=True, add_outputs=False):
        super().__init__()
        kwargs = {"groups": groups, "bias": bias, "batch_first": batch_first, "dropout": dropout,
                  "bidirectional": bidirectional}
        assert input_size % groups == 0
        assert hidden_size % groups == 0
        assert num_layers > 0
        self.input_size = input_size
        self.groups = groups
        self.num_layers = num_layers
        self.batch_first = batch_first
        self.hidden_size = hidden_size // groups
        self.bidirectional = bidirectional
        self.num_directions = 1
        if self.groups == 1:
            shuffle = False
        self.shuffle = shuffle
        self.add_outputs = add_outputs
        self.grus = nn.ModuleList()
        self.grus.append(GroupedGRULayer(input_size, hidden_size, **kwargs))
        for _ in range(1, num_layers):
            self.grus.append(GroupedGRULayer(hidden_size, hidden_size, **kwargs))

    def flatten_parameters(self):
        pass

    def get_h0(self, batch_size: int, device=None) -> Tensor:
        # the reference's forward calls get_h0(b, input.device) against a one-argument signature (:383-389,:397): the
        # evident intent (device optional) is what is implemented
        return torch.zeros((self.num_layers * self.groups * self.num_directions, batch_size, self.hidden_size), device=device)

    def forward(self, input: Tensor, state: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        dim0, dim1, _ = input.shape
        h = self.groups * self.num_directions
        if state is not None and state.shape[0] != self.num_layers * h:
            raise RuntimeError(f"GroupGRU: state has {state.shape[0]} rows, expected num_layers*groups = {self.num_layers * h}")
        output = None
        outstates = []
        for i, gru in enumerate(self.grus):
            input, s = gru(input, None if state is None else state[i * h:(i + 1) * h])                  # :402-404
            outstates.append(s)
            if self.shuffle and i < self.num_layers - 1:
                # a pure permutation of the feature axis: new[b*h + a] = old[a*g + b] (device copy, no arithmetic)
                input = input.view(dim0, dim1, -1, self.groups).transpose(2, 3).reshape(dim0, dim1, -1)
            if self.add_outputs:
                from ...nn_generic import add
                output = input if output is None else add(output, input.contiguous())
            else:
                output = input
        return output, torch.cat(outstates, dim=0)
