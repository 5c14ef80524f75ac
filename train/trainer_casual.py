"""Drop-in for `[trainer] path = "train.trainer_casual.Trainer"` (tools/train_stand.py:76)."""
from cruse_amd.train.trainer_casual import Trainer  # noqa: F401
