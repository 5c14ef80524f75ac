"""Namespace tools/train_stand.py:73-75 does getattr() on: `getattr(loss, name)(**args)`.

Every factory returns a callable usable stand-alone through autograd (HIP kernels underneath) that ALSO carries
`.cruse_loss = (engine loss name, kwargs)`: train.trainer_casual.Trainer reads it to select the fused loss of
cruse_amd.engine.TrainEngine, so `[loss_function] name/args` of the TOML is honoured.  l1_loss / mse_loss keep the
reference's aliases (train_base/loss.py:3-4): the Trainer recognises the torch.nn.L1Loss / MSELoss instance and runs the
fused waveform form (engine losses "l1" / "mse": iSTFT(mask * N) against the clean clip, reduction "mean")."""
import torch

l1_loss = torch.nn.L1Loss
mse_loss = torch.nn.MSELoss


def wo_male_loss(alpha=2.0, beta=1.0):
    """WO-MALE (loss_func/loss.py:121-148) fused with the mask application."""
    from cruse_amd.loss import masked_wo_male

    def fn(mask, noisy_real, noisy_imag, clean_mag):
        return masked_wo_male(mask, noisy_real, noisy_imag, clean_mag, alpha, beta)
    fn.cruse_loss = ("wo_male", dict(loss_alpha=float(alpha), loss_beta=float(beta)))
    return fn


def si_snr_loss():
    """train_base/loss.py:7-25 on the HIP path (the engine runs it through the iSTFT on the enhanced waveform)."""
    from cruse_amd.loss import si_snr_loss as _f
    fn = _f()
    fn.cruse_loss = ("si_snr", {})
    return fn


def sdnr_loss(snr=0.0, beta=20.0):
    """sdnr (loss_func/loss.py:151-175, vad == 1) with the mask as gain; snr / beta in dB."""
    from cruse_amd.loss import masked_sdnr

    def fn(mask, clean_real, clean_imag, noisy_real, noisy_imag):
        return masked_sdnr(mask, clean_real, clean_imag, noisy_real, noisy_imag, snr, beta)
    fn.cruse_loss = ("sdnr", dict(snr_db=float(snr), sdnr_beta_db=float(beta)))
    return fn
