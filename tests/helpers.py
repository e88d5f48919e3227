"""Shared test helpers: golden loading and oracle parameter construction (tests may use oracle/)."""
import glob
import json
import os

import numpy as np
import torch

from oracle import nasrec_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(pattern="*.npz"):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load_golden(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


def oracle_cfg(meta):
    ops = O.ops_config_lib[meta["config"]]
    return O.NetCfg(meta["num_blocks"], ops, meta["use_layernorm"], meta["activation"], fixed=(meta["mode"] == "fixed"))


def oracle_params(meta, dtype=torch.float64):
    """Name-seeded parameters with the key set / shapes the REFERENCE state_dict had (stored in the fixture)."""
    P = O.Params(dtype)
    for k, shp in meta["param_shapes"].items():
        P.get_or_create(k, shp)
    P.frozen = True
    return P


def proj_checksum(name, t):
    a = t.detach().double().cpu().numpy().reshape(-1)
    r = O.seeded_array("chk:" + name, a.shape)
    return float(np.dot(a, r)), float(np.linalg.norm(a))
