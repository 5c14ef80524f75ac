"""Re-export so the reference dotted path model.based_model.cust_conv resolves (train_base/utils.py:68-100)."""
from cruse_amd.model.based_model.cust_conv import *  # noqa: F401,F403
from cruse_amd.model.based_model.cust_conv import Conv2dNormAct, ConvTranspose2dNormAct, FreqUpsample, GroupedGRULayer, GroupGRU, convkxf  # noqa: F401
