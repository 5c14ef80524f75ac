"""Re-export so the reference dotted path train_base.acoustics.conv_stft resolves."""
from cruse_amd.acoustics.conv_stft import STFT  # noqa: F401
