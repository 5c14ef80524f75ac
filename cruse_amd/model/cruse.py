"""`model.cruse.CRUSE4MagAddSkipUpsample` (model/cruse.py:14).

The reference class is EMPTY (`pass`); its name spells the architecture -- CRUSE, 4 encoder / decoder levels, MAGnitude
input and mask, ADDitive SKIP connections, UPSAMPLE decoder -- and the reference ships every block it needs in
model/based_model/cust_conv.py: `Conv2dNormAct` (causal (2,3) encoder conv, frequency stride 2, BatchNorm, ReLU; :15-62),
`convkxf(mode="upsample")` (nearest `FreqUpsample` + Conv2d (1,3) instead of ConvTranspose2d; :114-174,177-184), and the
GGRU bottleneck of model/cruse_net.py:14-55.  DECISION (recorded in oracle.cruse_oracle_ext.CRUSE4MagAddSkipUpsample, which
this class mirrors child by child): the unet_2 topology (cruse_net.py:129-165 after repairs R2-R8) with exactly those blocks:

    e_k = Conv2dNormAct(ch[k-1], ch[k], (2,3), fstride=2)(e_{k-1})                k = 1..4     160 -> 80 -> 40 -> 20 -> 10 bins
    s_k = Conv2d(ch[k], ch[k], (1,3), padding=(0,1), bias=False)(e_k)             additive skips (cruse_net.py:143,153-156)
    u   = GGRU(hidden = ch[4]*10, groups)(e_4) + s_4
    d_k = convkxf(ch[k], ch[k-1], k=1, f=3, fstride=2, batch_norm=True, mode="upsample", depthwise=False)(d_{k+1}) + s_{k-1}
    mask = convkxf(ch[1], ch[0], ..., batch_norm=False, act=Sigmoid, mode="upsample", depthwise=False)(d_2)

The children are the general blocks (they own the parameters, so the state-dict keys are the composition's own), but
forward() does not call them: the model runs on the frame-major engine of unet_2 (cruse_net.unet2_forward / unet2_backward with
dec_mode="upsample": MFMA convs with the BatchNorm sums in their epilogues, the persistent GRU kernels, weight-gradient leaves
on the side streams) through one autograd node.  No CPU path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..nn_generic import HipConv2d
from .based_model.cust_conv import Conv2dNormAct, convkxf
from .cruse_net import DEFAULT_PREC, GGRU, unet2_backward, unet2_forward


def _unet2_names(laynum: int):
    """this model's parameter / buffer names -> the names cruse_net.unet2_forward reads (conv{k}, bn{k}, conv{k}_t, bn{k}_t)"""
    m = {}
    for k in range(1, laynum + 1):
        for f in ("weight", "bias"):
            m[f"enc{k}.1.{f}"] = f"conv{k}.{f}"
            m[f"dec{k}.sconv.{f}"] = f"conv{k}_t.{f}"
        for f in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            m[f"enc{k}.2.{f}"] = f"bn{k}.{f}"
            m[f"dec{k}.norm.{f}"] = f"bn{k}_t.{f}"
    return m


class _Cruse4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, names, *params):
        alias = mod._alias
        P = {alias.get(n, n): p for n, p in zip(names, params)}
        Bf = {alias.get(n, n): b for n, b in mod.named_buffers()}
        need = x.requires_grad or any(p.requires_grad for p in params)
        mask, c = unet2_forward(x, P, Bf, mod.ch, mod.gru.groups, mod.gru.precision, mod.training, save=need, dec_mode="upsample")
        ctx.c, ctx.P, ctx.names, ctx.alias, ctx.need_dx = c, P, names, alias, x.requires_grad
        return mask

    @staticmethod
    def backward(ctx, dmask):
        P, c = ctx.P, ctx.c
        G = {n: torch.zeros_like(p) for n, p in P.items()}
        B, T, F0 = c["B"], c["T"], c["F"][0]
        dlogit = ops.sigmoid_bwd(dmask.contiguous().view(B, T, 1, F0), c["mask"])
        dx = unet2_backward(c, dlogit, P, G, need_dx=ctx.need_dx)
        return (dx.view(B, 1, T, F0) if dx is not None else None, None, None) + tuple(G[ctx.alias.get(n, n)] for n in ctx.names)


class CRUSE4MagAddSkipUpsample(nn.Module):
    def __init__(self, in_feat=161, ch=(1, 8, 16, 32, 64), rnn_groups=1, precision: str = DEFAULT_PREC):
        super().__init__()
        self.laynum = len(ch) - 1
        self.ch = tuple(ch)
        self.f_net = in_feat // 2 ** self.laynum * 2 ** self.laynum
        hidden = in_feat // 2 ** self.laynum * ch[-1]
        for k in range(1, self.laynum + 1):
            setattr(self, f"enc{k}", Conv2dNormAct(ch[k - 1], ch[k], (2, 3), fstride=2))
            setattr(self, f"skip_connect_{k}", HipConv2d(ch[k], ch[k], (1, 3), padding=(0, 1), bias=False))
            last = k == 1
            setattr(self, f"dec{k}", convkxf(ch[k], ch[k - 1], k=1, f=3, fstride=2, batch_norm=not last,
                                             act=nn.Sigmoid() if last else nn.ReLU(), mode="upsample", depthwise=False))
        self.gru = GGRU(hidden_size=hidden, groups=rnn_groups, precision=precision)
        self._alias = _unet2_names(self.laynum)

    def forward(self, x):
        """x [B,1,T,F] magnitude (F = 160 for in_feat 161) -> mask [B,1,T,F]."""
        if x.dim() != 4 or x.shape[1] != self.ch[0] or x.shape[-1] != self.f_net:
            raise RuntimeError(f"CRUSE4MagAddSkipUpsample expects [B,{self.ch[0]},T,{self.f_net}], got {tuple(x.shape)}")
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        return _Cruse4Fn.apply(x.contiguous().float(), self, names, *params)
