"""Re-export so the reference dotted path train_base.acoustics.mask resolves."""
from cruse_amd.acoustics.mask import *  # noqa: F401,F403
from cruse_amd.acoustics.mask import build_complex_ideal_ratio_mask, build_ideal_ratio_mask, complex_mul, compress_cIRM, decompress_cIRM  # noqa: F401
