"""The training step the reference never wrote (train/trainer_casual.py is empty):

    STFT(noisy), STFT(clean) -> unet_2 -> mask * spectrum -> WO-MALE -> backward
    -> gradient all-reduce (RCCL) -> Adam

run on one MI355X per process.  Parameters, gradients and Adam moments live in flat,
64-float-aligned buffers (one all-reduce, one fused Adam launch); the forward+backward
kernel sequence is recorded once into a HIP graph and replayed.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .model.cruse_net import unet2_backward, unet2_forward, unet_2

ALIGN = 64  # floats


def unused_parameter(name: str) -> bool:
    """`fc` (cruse_net.py:146) and `bn1_t` (:142 after R3) never receive gradients."""
    return name.startswith("fc.") or name.startswith("bn1_t.")


class FlatParams:
    """Views of a model's trainable parameters inside one contiguous buffer (plus grads / Adam state).

    Works on any device (the CPU/gloo tests exercise the layout and the all-reduce)."""

    def __init__(self, model: nn.Module, skip=unused_parameter):
        named = [(n, p) for n, p in model.named_parameters() if not skip(n)]
        self.names = [n for n, _ in named]
        self.offsets: Dict[str, int] = {}
        off = 0
        for n, p in named:
            self.offsets[n] = off
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        dev = named[0][1].device
        self.params = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for n, p in named:
                o, k = self.offsets[n], p.numel()
                view = self.params[o:o + k].view(p.shape)
                view.copy_(p)
                p.data = view                       # the module now reads the flat buffer
                self.P[n] = view
                self.G[n] = self.grads[o:o + k].view(p.shape)
        # parameters outside the flat buffer (never trained) are still needed by name
        for n, p in model.named_parameters():
            if n not in self.P:
                self.P[n] = p.data

    def broadcast(self, src: int = 0) -> None:
        """identical initial weights on all ranks (cf. loss_func/distrib.py:57-72)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.params, src)

    def all_reduce_grads(self) -> None:
        """sum over ranks; the 1/world_size factor is folded into the Adam kernel."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM)

    def attach_grads(self, model: nn.Module) -> None:
        """expose the flat gradients as param.grad (checkpoint / clip_grad_norm_ compatibility)."""
        for n, p in model.named_parameters():
            if n in self.G:
                p.grad = self.G[n]


class TrainEngine:
    def __init__(self, model: unet_2, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 n_fft=320, hop=160, precision: Optional[str] = None, use_graph: bool = True,
                 loss_alpha=2.0, loss_beta=1.0, loss: str = "wo_male"):
        if not torch.cuda.is_available():
            raise RuntimeError("cruse_amd.TrainEngine needs a HIP device (there is no CPU path)")
        self.model = model
        self.n_fft, self.hop = n_fft, hop
        self.f_net = (n_fft // 2 + 1) // 2 * 2
        self.f_stft = n_fft // 2 + 1
        if precision is not None:
            model.set_precision(precision)
        self.prec = model.precision
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.loss_alpha, self.loss_beta = loss_alpha, loss_beta
        if loss not in ("wo_male", "si_snr"):
            raise ValueError(f"unknown loss {loss!r} (wo_male | si_snr)")
        self.loss = loss
        self.flat = FlatParams(model)
        self.flat.broadcast(0)
        self.Bf = dict(model.named_buffers())
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.step_count = 0
        self.use_graph = use_graph
        self._graph = None
        self._static = None
        self._shape = None

    # -- one forward + loss + backward, gradients left in flat.grads --------------------
    def _fwd_bwd(self, noisy: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
        B, L = noisy.shape
        T = ops.stft_frames(L, self.hop)
        nre, nim, mag = ops.stft(noisy, self.n_fft, self.hop, mag_bins=self.f_net, mag_eps=1e-8)
        cmag = None
        if self.loss == "wo_male":
            _, _, cmag = ops.stft(clean, self.n_fft, self.hop, want_ri=False, mag_bins=self.f_stft, mag_eps=0.0)
        mask, ctx = unet2_forward(mag.view(B, 1, T, self.f_net), self.flat.P, self.Bf, self.model.ch,
                                  self.model.rnn_groups, self.prec, training=True)
        self._last_mask = mask
        if self.loss == "wo_male":
            loss_sum, _, dlogit, _, _ = ops.mask_loss(mask, nre, nim, cmag, B * T, self.f_net, self.f_stft,
                                                      self.loss_alpha, self.loss_beta, want_dlogit=True)
            self._norm = float(B * T * self.f_stft)
        else:
            # waveform in -> waveform loss: est = iSTFT(mask * N); SI-SNR(est, clean) and back through both
            ere, eim = ops.mask_apply(mask, nre, nim, B * T, self.f_net, self.f_stft)
            est = ops.istft(ere.view(B, T, self.f_stft), eim.view(B, T, self.f_stft), self.n_fft, self.hop, L)
            loss_sum, coef = ops.sisnr_fwd(est, clean)
            dwave = ops.sisnr_bwd(est, clean, coef)
            dre, dim = ops.istft_bwd(dwave, T, self.n_fft, self.hop)
            dlogit = ops.mask_apply_bwd(dre, dim, nre, nim, mask, B * T, self.f_net, self.f_stft)
            self._norm = 1.0
        self.flat.grads.zero_()
        unet2_backward(ctx, dlogit.view(B, T, 1, self.f_net), self.flat.P, self.flat.G)
        return loss_sum

    def _capture(self, noisy, clean):
        self._static = (noisy.clone(), clean.clone())
        # the warm-up run below must leave no trace: BatchNorm running statistics and counters are restored
        saved = {k: v.clone() for k, v in self.Bf.items()}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):           # warm-up on a side stream (allocations, LDS attributes)
            self._fwd_bwd(*self._static)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for k, v in saved.items():
            self.Bf[k].copy_(v)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._static_loss = self._fwd_bwd(*self._static)
        self._graph = g
        self._shape = tuple(noisy.shape)

    def step(self, noisy: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
        """One optimizer step; returns the (device, f64) loss sum -- divide by .loss_norm for the loss."""
        if self.use_graph:
            if self._graph is None or self._shape != tuple(noisy.shape):
                self._capture(noisy, clean)
            self._static[0].copy_(noisy)
            self._static[1].copy_(clean)
            self._graph.replay()
            loss_sum = self._static_loss
        else:
            loss_sum = self._fwd_bwd(noisy, clean)
        self.flat.all_reduce_grads()
        self.step_count += 1
        ops.adam_step(self.flat.params, self.flat.grads, self.flat.exp_avg, self.flat.exp_avg_sq, self.lr,
                      self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, 1.0 / self.world)
        return loss_sum

    @property
    def loss_norm(self) -> float:
        return self._norm

    def loss_value(self, loss_sum: torch.Tensor) -> float:
        return float(loss_sum.item()) / self._norm
