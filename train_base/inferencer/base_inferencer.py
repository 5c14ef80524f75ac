"""Reference import path (train_base/inferencer/base_inferencer.py) -> the HIP inferencer."""
from cruse_amd.inferencer.base_inferencer import Inferencer  # noqa: F401
