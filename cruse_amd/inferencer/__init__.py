from .base_inferencer import Inferencer  # noqa: F401
