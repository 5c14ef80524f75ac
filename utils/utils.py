"""Re-export so the reference dotted path utils.utils.PreProcess resolves (utils/utils.py:365-455)."""
from cruse_amd.acoustics.preprocess import PreProcess  # noqa: F401
