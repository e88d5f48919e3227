from nasrec_amd.utils.data_pipes import *  # noqa: F401,F403
from nasrec_amd.utils.data_pipes import get_avazu_kaggle_pipes, get_criteo_kaggle_pipes, get_kdd_kaggle_pipes, make_loaders  # noqa: F401
