from nasrec_amd.utils.train_utils import *  # noqa: F401,F403
from nasrec_amd.utils.train_utils import (accuracy, get_l2_loss, get_model_flops_and_params, get_model_latency, init_weights,  # noqa: F401
                                          test_one_epoch, train_and_test_one_epoch, warmup_model, warmup_supernet_model)
