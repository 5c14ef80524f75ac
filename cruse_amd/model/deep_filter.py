"""MI355X-native `model.deep_filter.DeepFilter` (model/deep_filter.py:15-41, BASELINE config 4).

forward(inputs=[re, im], filters=[re, im]) with every tensor [B,F,T] -> cat([out_r, out_i], dim=1)
([B,2F,T]).  The `kernel` buffer of the reference (identity unfold kernel, :22-26 with the ctor repair
`[t_width*f_width, 1, f_width, t_width]`) is kept for state-dict compatibility; the op itself is the fused
box-sum-of-products kernel cruse_deepfilter_fwd / _bwd.  Imaginary part: xr*hi + xi*hr (the reference's :38
writes xr*hi twice; SURVEY.md 8a a15 -- decision: the mathematically correct product)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class _DeepFilterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xr, xi, hr, hi, f_dim, t_dim):
        xr, xi, hr, hi = (t.contiguous() for t in (xr, xi, hr, hi))
        ctx.save_for_backward(xr, xi, hr, hi)
        ctx.dims = (f_dim, t_dim)
        o_r, o_i = ops.deepfilter_fwd(xr, xi, hr, hi, f_dim, t_dim)
        return torch.cat([o_r, o_i], dim=1)

    @staticmethod
    def backward(ctx, dout):
        xr, xi, hr, hi = ctx.saved_tensors
        F = xr.shape[1]
        dor, doi = dout[:, :F].contiguous(), dout[:, F:].contiguous()
        dxr, dxi, dhr, dhi = ops.deepfilter_bwd(dor, doi, xr, xi, hr, hi, *ctx.dims)
        return dxr, dxi, dhr, dhi, None, None


class DeepFilter(nn.Module):
    def __init__(self, t_dim, f_dim):
        super().__init__()
        self.t_dim, self.f_dim = t_dim, f_dim
        t_width, f_width = t_dim * 2 + 1, f_dim * 2 + 1
        self.register_buffer("kernel", torch.eye(t_width * f_width).reshape(t_width * f_width, 1, f_width, t_width))

    def forward(self, inputs, filters):
        xr, xi = inputs
        hr, hi = filters
        if not (xr.shape == xi.shape == hr.shape == hi.shape) or xr.dim() != 3:
            raise RuntimeError(f"DeepFilter expects four [B,F,T] tensors, got {[tuple(t.shape) for t in (xr, xi, hr, hi)]}")
        return _DeepFilterFn.apply(xr, xi, hr, hi, self.f_dim, self.t_dim)
