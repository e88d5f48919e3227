"""Per-operator golden cases (tests/golden/ops_*.npz) and the glue that runs one case on the ORACLE (CPU) — shared by
tests/test_ops_golden.py (oracle vs reference vectors) and tests/test_operator_api_gpu.py (HIP operators vs the same)."""
import json

import numpy as np
import torch

from helpers import golden_files
from oracle import nasrec_oracle as O


def all_cases():
    out = {}
    for path in golden_files("ops_*.npz"):
        z = np.load(path, allow_pickle=False)
        metas = json.loads(str(z["meta"]))
        for case, meta in metas.items():
            out[case] = (z, meta)
    return out


def case_inputs(case, z, meta, dtype, device="cpu", requires_grad=True):
    xs = []
    for i in range(len(meta["in_shapes"])):
        x = torch.tensor(z["%s/in%d" % (case, i)], dtype=dtype, device=device)
        xs.append(x.requires_grad_(requires_grad))
    return xs


def oracle_run(case, z, meta, dtype, frozen=True, warm=False):
    """run the oracle's restatement of the case's operator -> (Params, (inputs, [outputs]))"""
    P = O.Params(dtype)
    if frozen:
        for k, shp in meta["param_shapes"].items():
            P.get_or_create("op." + k, shp)
        P.frozen = True
        for t in P.values():
            t.requires_grad_(True)
    xs = case_inputs(case, z, meta, dtype)
    cls, dims = meta["cls"], meta["dims_in_use"]
    fixed = meta["fixed"]
    kw = meta.get("kwargs", {})
    ln = kw.get("use_layernorm")
    md = kw.get("max_dims_or_dims")
    act = kw.get("activation", "relu")
    if cls == "ElasticLinear":
        out = O.elastic_linear(P, "op", xs[0], dims, md, ln, act, fixed)
    elif cls == "ElasticLinear3D":
        out = O.elastic_linear3d(P, "op", xs[0], dims, md, ln, act, fixed)
    elif cls == "DotProduct":
        out = O.dot_product(P, "op", xs[0], xs[1], dims, md, ln, 16, fixed)
    elif cls == "Sum":
        out = O.sum_node(P, "op", xs[0], xs[1], dims, md, ln, fixed)
    elif cls == "SigmoidGating":
        out = O.sigmoid_gating(P, "op", xs[0], xs[1], dims, md, ln, fixed)
    elif cls == "Transformer":
        out = O.transformer(P, "op", xs[0], dims, md, ln, 16, fixed)
    elif cls == "FactorizationMachine3D":
        out = O.fm3d(P, "op", xs[0], dims, md, ln, fixed)
    elif cls == "Zeros2D":  # modules.py:238-270
        out = torch.zeros(xs[0].shape[0], dims if fixed else md, dtype=dtype)
    elif cls == "Zeros3D":  # modules.py:691-718
        out = torch.zeros(xs[0].shape[0], md, xs[0].shape[2], dtype=dtype)
    elif cls == "SuperNetBlock":
        ops = O.ops_config_lib[meta["space"]]
        cfg = O.NetCfg(1, ops, meta["use_layernorm"], meta["activation"], fixed=fixed)
        choice = meta["choice"]
        if warm and not fixed:  # the reference warms a weight-sharing block up on the full path (train_utils.py:413-433)
            choice = {"active_nodes": list(range(ops["num_nodes"])), "dense_in_dims": max(ops["dense_node_dims"]),
                      "sparse_in_dims": max(ops["sparse_node_dims"]), "dense_sparse_interact": 1, "deep_fm": 1}
        out = O.block_forward(P, "op", cfg, ops, xs, choice)
    else:
        raise KeyError(cls)
    outs = list(out) if isinstance(out, (tuple, list)) else [out]
    return P, (xs, outs)
