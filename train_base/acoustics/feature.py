"""Drop-in for `train_base.acoustics.feature.stft / istft` (feature.py:10-61) on the HIP path."""
from cruse_amd.acoustics.feature import istft, pre_stft, stft  # noqa: F401
