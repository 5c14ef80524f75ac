import torch


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    """global relative L2 error ||a-b|| / ||b|| (SURVEY 8d parity gate)."""
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.is_complex() or b.is_complex():                      # complex spectra: as (re, im) pairs
        a, b = torch.view_as_real(a.resolve_conj().to(torch.complex128)), torch.view_as_real(b.resolve_conj().to(torch.complex128))
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def t(x, device="cuda"):
    import numpy as np
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device).float().contiguous()
