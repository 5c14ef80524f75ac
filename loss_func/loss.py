"""Re-export so the reference dotted path loss_func.loss resolves."""
from cruse_amd.loss_func import c_rmse, loss_func, rmse, sdnr, sisnr, wo_male  # noqa: F401
