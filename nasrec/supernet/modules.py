from nasrec_amd.supernet.modules import *  # noqa: F401,F403
from nasrec_amd.supernet.modules import (LN_INIT, NUM_MHA_HEADS, CleverMaskGenerator, CleverZeroTensorGenerator, DotProduct,  # noqa: F401
                                         ElasticLinear, ElasticLinear3D, FactorizationMachine3D, LazySelfLinear, SigmoidGating,
                                         Sum, Transformer, Zeros2D, Zeros3D, apply_activation_fn, flags)
