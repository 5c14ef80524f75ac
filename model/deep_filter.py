"""Drop-in for the reference's dotted path `model.deep_filter.DeepFilter`."""
from cruse_amd.model.deep_filter import DeepFilter  # noqa: F401
