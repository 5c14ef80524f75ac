"""Namespace tools/train_stand.py:73-75 does getattr() on.  `wo_male_loss` returns the fused
HIP mask+WO-MALE criterion; l1_loss / mse_loss keep the reference's aliases (train_base/loss.py:3-4)."""
import torch

l1_loss = torch.nn.L1Loss
mse_loss = torch.nn.MSELoss


def wo_male_loss(alpha=2.0, beta=1.0):
    from cruse_amd.loss import masked_wo_male

    def fn(mask, noisy_real, noisy_imag, clean_mag):
        return masked_wo_male(mask, noisy_real, noisy_imag, clean_mag, alpha, beta)
    return fn


def si_snr_loss():
    """train_base/loss.py:7-25 on the HIP path."""
    from cruse_amd.loss import si_snr_loss as _f
    return _f()
