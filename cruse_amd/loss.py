"""Magnitude-mask application + weighted spectral loss on the HIP path.

`masked_wo_male` fuses PreProcess.masking "mag_mapping" (utils/utils.py:418-420) with
WO-MALE (loss_func/loss.py:121-148, called as wo_male(labels, inputs, noisy) from
loss_func.loss at :24,30) and its gradient wrt the mask in one kernel
(cruse_mask_loss_fwd).
"""
from __future__ import annotations

import torch

from . import ops


class _MaskedWoMaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, nre, nim, cmag, alpha, beta):
        B, _, T, Fn = mask.shape
        Fs = nre.shape[-1]
        rows = B * T
        need = mask.requires_grad
        loss_sum, dmask, _, _, _ = ops.mask_loss(mask.contiguous(), nre.contiguous(), nim.contiguous(),
                                                 cmag.contiguous(), rows, Fn, Fs, alpha, beta, want_dmask=need)
        ctx.dmask = dmask
        ctx.shape = mask.shape
        return (loss_sum / float(rows * Fs)).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        # chain rule with the upstream scalar (normally 1.0); the engine path (cruse_amd.engine)
        # consumes the kernel's dlogit directly and never comes through here
        return ctx.dmask.view(ctx.shape) * g, None, None, None, None, None


def masked_wo_male(mask, noisy_real, noisy_imag, clean_mag, alpha=2.0, beta=1.0):
    """mask [B,1,T,Fn]; noisy_real/imag [B,T,Fs] (or [B,1,T,Fs]); clean_mag [B,T,Fs] = |STFT(clean)|.
    Bins Fn..Fs-1 of the estimate are zero.  Returns the scalar WO-MALE loss."""
    if mask.dim() != 4:
        raise RuntimeError(f"masked_wo_male: mask must be [B,1,T,F], got {tuple(mask.shape)}")
    Fs = noisy_real.shape[-1]
    B, _, T, Fn = mask.shape
    for name, t in (("noisy_real", noisy_real), ("noisy_imag", noisy_imag), ("clean_mag", clean_mag)):
        if t.numel() != B * T * Fs:
            raise RuntimeError(f"Dimension mismatch when calculate wo-male, {name} {tuple(t.shape)} vs mask {tuple(mask.shape)}")
    return _MaskedWoMaleFn.apply(mask, noisy_real, noisy_imag, clean_mag, float(alpha), float(beta))


def enhanced_spectrum(mask, noisy_real, noisy_imag):
    """PreProcess.masking (utils/utils.py:417-433): -> [B,T,Fs,2]; bins >= Fn are zero."""
    B, _, T, Fn = mask.shape
    Fs = noisy_real.shape[-1]
    rows = B * T
    dummy = torch.ones(rows, Fs, device=mask.device, dtype=torch.float32)
    _, _, _, er, ei = ops.mask_loss(mask.contiguous(), noisy_real.contiguous(), noisy_imag.contiguous(), dummy,
                                    rows, Fn, Fs, want_est=True)
    return torch.stack([er.view(B, T, Fs), ei.view(B, T, Fs)], dim=-1)


class _SiSnrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s, eps):
        x = x.contiguous(); s = s.contiguous()
        loss, coef = ops.sisnr_fwd(x, s, eps)
        ctx.save_for_backward(x, s, coef)
        return loss.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        x, s, coef = ctx.saved_tensors
        return ops.sisnr_bwd(x, s, coef) * g, None, None


def si_snr_loss():
    """train_base/loss.py:7-25: returns si_snr(x, s, eps=1e-8) on [B,L] waveforms (HIP kernels)."""
    def si_snr(x, s, eps=1e-8):
        if x.shape != s.shape:
            raise RuntimeError(f"Dimension mismatch when calculate si_snr, {x.shape} vs {s.shape}")
        return _SiSnrFn.apply(x, s, float(eps))
    return si_snr


class _MaskedSdnrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, cre, cim, nre, nim, snr_db, beta_db):
        B, _, T, Fn = mask.shape
        Fs = nre.shape[-1]
        rows = B * T
        need = mask.requires_grad
        loss_sum, dmask, _ = ops.mask_sdnr(mask.contiguous(), cre.contiguous(), cim.contiguous(), nre.contiguous(),
                                           nim.contiguous(), rows, Fn, Fs, B, snr_db, beta_db, want_dmask=need)
        ctx.dmask, ctx.shape = dmask, mask.shape
        return (loss_sum / float(B * Fs)).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        return ctx.dmask.view(ctx.shape) * g, None, None, None, None, None, None


def masked_sdnr(mask, clean_real, clean_imag, noisy_real, noisy_imag, snr_db=0.0, beta_db=20.0):
    """sdnr(ref_clean, est_g = mask, ref_noise = noisy - clean, snr) of loss_func/loss.py:151-175 (vad == 1).
    mask [B,1,T,Fn]; spectra [B,T,Fs] (or [B,1,T,Fs]); gain bins Fn..Fs-1 are zero (R8)."""
    if mask.dim() != 4:
        raise RuntimeError(f"masked_sdnr: mask must be [B,1,T,F], got {tuple(mask.shape)}")
    B, _, T, Fn = mask.shape
    Fs = noisy_real.shape[-1]
    for name, t in (("clean_real", clean_real), ("clean_imag", clean_imag), ("noisy_real", noisy_real),
                    ("noisy_imag", noisy_imag)):
        if t.numel() != B * T * Fs:
            raise RuntimeError(f"Dimension mismatch when calculate sdnr, {name} {tuple(t.shape)} vs mask {tuple(mask.shape)}")
    return _MaskedSdnrFn.apply(mask, clean_real, clean_imag, noisy_real, noisy_imag, float(snr_db), float(beta_db))
