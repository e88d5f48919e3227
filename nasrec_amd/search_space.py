"""Search-space tables of the reference (nasrec/supernet/supernet.py:116-207), restated as data."""

DENSE_NODE_DIMS = [16, 32, 64, 128, 256, 512, 768, 1024]
SPARSE_NODE_DIMS = [16, 32, 48, 64]

ops_config_lib = {
    "xlarge": {
        "num_nodes": 6,
        "node_names": ["linear-2d", "dot-product", "sigmoid-gating", "sum", "transformer", "linear-3d"],
        "dense_node_dims": DENSE_NODE_DIMS, "sparse_node_dims": SPARSE_NODE_DIMS,
        "dense_nodes": [0, 1, 2, 3], "sparse_nodes": [4, 5], "zero_nodes": [],
    },
    "xlarge-zeros": {
        "num_nodes": 8,
        "node_names": ["linear-2d", "dot-product", "sigmoid-gating", "sum", "zeros-2d", "transformer", "zeros-3d", "linear-3d"],
        "dense_node_dims": DENSE_NODE_DIMS, "sparse_node_dims": SPARSE_NODE_DIMS,
        "dense_nodes": [0, 1, 2, 3, 4], "sparse_nodes": [5, 6, 7], "zero_nodes": [4, 6],
    },
    "autoctr": {
        "num_nodes": 3,
        "node_names": ["linear-2d", "dot-product", "linear-3d"],
        "dense_node_dims": DENSE_NODE_DIMS, "sparse_node_dims": SPARSE_NODE_DIMS,
        "dense_nodes": [0, 1], "sparse_nodes": [2], "zero_nodes": [],
    },
}

path_sampling_strategy_lib = {
    "default": {"macro": "any-path", "micro": "single-path"},
    "single-path": {"macro": "single-path", "micro": "single-path"},
    "any-path": {"macro": "any-path", "micro": "any-path"},
    "full-path": {"macro": "full-path", "micro": "full-path"},
    "fixed-path": {"macro": "fixed-path", "micro": "fixed-path"},
    "evo-2shot-path": {"macro": "evo-2shot-path", "micro": "evo-2shot-path"},
}
