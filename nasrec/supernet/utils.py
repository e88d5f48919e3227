from nasrec_amd.supernet.utils import anypath_choice_fn, assert_valid_ops_config  # noqa: F401
