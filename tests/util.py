import torch


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    """global relative L2 error ||a-b|| / ||b|| (SURVEY 8d parity gate)."""
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def t(x, device="cuda"):
    import numpy as np
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device).float().contiguous()
