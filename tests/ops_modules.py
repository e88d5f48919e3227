"""Build the build's own operator module for a per-operator golden case (tests/golden/ops_*.npz)."""
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.supernet import modules as M
from nasrec_amd.supernet.supernet import SuperNetBlock


def build_module(meta):
    cls = meta["cls"]
    if cls == "SuperNetBlock":
        ops = ops_config_lib[meta["space"]]
        fixed = meta["fixed"]
        return SuperNetBlock(ops, meta["use_layernorm"], max(ops["dense_node_dims"]), max(ops["sparse_node_dims"]), 16,
                             activation=meta["activation"], path_sampling_strategy="fixed-path" if fixed else "full-path", fixed=fixed,
                             fixed_micro_choice=meta["choice"] if fixed else None)
    if cls == "Zeros3D":
        return M.Zeros3D(**meta["kwargs"])
    return getattr(M, cls)(fixed=meta["fixed"], **meta["kwargs"])
