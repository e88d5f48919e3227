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
    return O.NetCfg(meta["num_blocks"], ops, meta["use_layernorm"], meta["activation"], fixed=(meta["mode"] == "fixed"),
                    last_n_blocks_out=meta.get("last_n_blocks_out", 1))


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


def check_trajectory(z, meta, losses, norms, params, rel_params, rel_delta, rel_loss):
    """Every step of the clipped-Adagrad trajectory against the reference's fp64 run (tests/golden/make_golden.py run_case):
    per-step loss and pre-clip gradient norm; after the last step, for EVERY parameter both the projection checksum
    <p, r_name> and the L2 norm within rel_params * |p_ref|, and the same two numbers of the update p_after - p_seeded within
    rel_delta * max(|update_ref|, 1 % of a full-size Adagrad update) — the scale the fixture's own `ref_fp32_drift` is
    quoted on (the reference's fp32 run drifts from its fp64 run by <= 9e-6 / 1.8e-3 on these two scales)."""
    ref_l, ref_n = z["step_losses"], z["step_gradnorms"]
    assert len(losses) == meta["n_steps"] == len(ref_l)
    for t in range(meta["n_steps"]):
        assert abs(losses[t] - ref_l[t]) <= rel_loss * max(1.0, abs(ref_l[t])), ("loss", t, losses[t], ref_l[t])
        assert abs(norms[t] - ref_n[t]) <= rel_loss * max(1.0, ref_n[t]), ("grad norm", t, norms[t], ref_n[t])
    lr, n_steps = meta["lr"], meta["n_steps"]
    bad = []
    for k, (dot, nrm) in meta["params_after"].items():
        d, n = proj_checksum(k, params[k])
        tol = rel_params * max(nrm, 1e-12)
        if abs(d - dot) > tol or abs(n - nrm) > tol:
            bad.append((k, "param", d - dot, n - nrm, tol))
        ddot, dnrm = meta["params_delta"][k]
        seeded = torch.tensor(O.seeded_param(k, meta["param_shapes"][k]), dtype=torch.float64)
        dd, dn = proj_checksum(k, params[k].detach().double().cpu().reshape(seeded.shape) - seeded)
        scale = max(dnrm, 1e-2 * lr * n_steps * float(np.sqrt(seeded.numel())))
        # the engine holds fp32 parameters: p32 - p_seeded64 carries the fp32 rounding of the seeded value itself
        slack = 6e-8 * max(nrm, 1e-12)
        if abs(dd - ddot) > rel_delta * scale + slack or abs(dn - dnrm) > rel_delta * scale + slack:
            bad.append((k, "update", dd - ddot, dn - dnrm, rel_delta * scale))
    assert not bad, "%d parameter checks failed after %d steps: %s" % (len(bad), n_steps, bad[:8])
