"""Re-export of the on-GPU `SynDataset.snr_mix` (dataset/dataset.py:236-264)."""
from cruse_amd.data import snr_mix  # noqa: F401


class SynDataset:
    """Only the mixing step of the reference class is on this path; it is a staticmethod there (dataset.py:235)."""
    snr_mix = staticmethod(snr_mix)
