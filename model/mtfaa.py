"""Re-export so the reference dotted path model.mtfaa resolves (train_base/utils.py:68-100)."""
from cruse_amd.model.mtfaa import STFT, ComplexConv2d, ComplexLinearProjection, PhaseEncoder, TFCM, TFCM_Block, complex_cat  # noqa: F401
