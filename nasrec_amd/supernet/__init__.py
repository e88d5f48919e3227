from .supernet import SuperNet, SuperNetBlock, ops_config_lib, path_sampling_strategy_lib  # noqa: F401
