"""Sampling helpers of the search space (reference: nasrec/supernet/utils.py).  Both samplers draw from the global
``np.random`` stream: the RNG call order defines which path every training step takes."""
import numpy as np


def _get_random_choice_vanilla(num_items, max_items=4):
    """uniform over 1..min(num_items, max_items) (utils.py:21-28)"""
    return np.random.choice(min(num_items, max_items)) + 1


def _get_binomial_random_choice_with_expectation(num_items, p=0.5, max_items=4):
    """1 + Binomial(min(num_items, max_items) - 1, p) (utils.py:31-36)"""
    return 1 + np.random.binomial(min(num_items - 1, max_items - 1), p)


anypath_choice_fn = {
    "uniform": lambda num_items: _get_random_choice_vanilla(num_items, max_items=4),
    "binomial-0.5": lambda num_items: _get_binomial_random_choice_with_expectation(num_items, p=0.5, max_items=4),
}


def assert_valid_ops_config(ops_config):
    """utils.py:46-61: every search space must name exactly num_nodes nodes"""
    for key, cfg in ops_config.items():
        for c in (cfg if isinstance(cfg, list) else [cfg]):
            assert c["num_nodes"] == len(c["node_names"]), ValueError(
                "Number of nodes per config should be equivalent to the number of modules (node names) per config.")
