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

Layout [B,C,T,F] throughout (the blocks are the general NCHW ones: generic.hip; the nearest upsampling is folded into the
following convolution's gather index, never materialised); the bottleneck runs the persistent GRU kernels.  No CPU path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..nn_generic import HipConv2d, add
from .based_model.cust_conv import Conv2dNormAct, convkxf
from .cruse_net import DEFAULT_PREC, GGRU


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

    def forward(self, x):
        """x [B,1,T,F] magnitude (F = 160 for in_feat 161) -> mask [B,1,T,F]."""
        if x.dim() != 4 or x.shape[1] != self.ch[0] or x.shape[-1] != self.f_net:
            raise RuntimeError(f"CRUSE4MagAddSkipUpsample expects [B,{self.ch[0]},T,{self.f_net}], got {tuple(x.shape)}")
        e, skips = x, []
        for k in range(1, self.laynum + 1):
            e = getattr(self, f"enc{k}")(e)
            skips.append(getattr(self, f"skip_connect_{k}")(e))
        d = add(self.gru(e), skips[-1])
        for k in range(self.laynum, 1, -1):
            d = add(getattr(self, f"dec{k}")(d), skips[k - 2])
        return self.dec1(d)
