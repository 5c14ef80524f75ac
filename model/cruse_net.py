"""Drop-in for the reference's dotted path `model.cruse_net.unet_2` / `model.cruse_net.GGRU`
(initialize_module, train_base/utils.py:68-100): re-exports the MI355X-native classes."""
from cruse_amd.model.cruse_net import GGRU, unet_2  # noqa: F401
