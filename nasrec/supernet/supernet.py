from nasrec_amd.supernet.supernet import *  # noqa: F401,F403
from nasrec_amd.supernet.supernet import (DS_INTERACT_NUM_SPLITS, SuperNet, SuperNetBlock, ops_config_lib,  # noqa: F401
                                          path_sampling_strategy_lib)
