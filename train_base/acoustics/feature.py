"""Drop-in for `train_base.acoustics.feature` (feature.py:10-61 stft / istft, :272-398 CustomSTFT / CustomISTFT) on the HIP path."""
from cruse_amd.acoustics.feature import (CustomISTFT, CustomSTFT, CustomSTFTBase, init_stft_kernel, istft,  # noqa: F401
                                         pre_stft, stft)
