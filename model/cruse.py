"""Drop-in for the reference's dotted path `model.cruse.CRUSE4MagAddSkipUpsample` (model/cruse.py:14)."""
from cruse_amd.model.cruse import CRUSE4MagAddSkipUpsample  # noqa: F401
