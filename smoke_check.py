"""One small invocation of the hot path on cuda:0, checked against the CPU oracle (oracle/ is the checker only).
Lives OUTSIDE the product package: nothing under nasrec_amd/ imports oracle/."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))

# Criteo best-1shot sub-network (the choice published in nasrec/configs/criteo/ea_criteo_kaggle_xlarge_best_1shot.json,
# carried by the golden fixture's metadata)
FIXTURE = os.path.join(ROOT, "tests", "golden", "fixed_criteo_xlarge.npz")


def run_smoke():
    from nasrec_amd import plan as P
    from nasrec_amd.engine import SupernetEngine
    from nasrec_amd.search_space import ops_config_lib
    from oracle import nasrec_oracle as O  # checker

    z = np.load(FIXTURE, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    cfg = P.NetConfig(meta["num_blocks"], ops_config_lib[meta["config"]], meta["use_layernorm"], meta["activation"], fixed=True)
    eng = SupernetEngine(cfg, 13, 26, meta["tables"], device="cuda:0", warm_choice=meta["choice"])
    eng.load_params({k: O.seeded_param(k, s) for k, s in meta["param_shapes"].items()})
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda()
    # oracle: fp64 forward + one clipped Adagrad step
    ocfg = O.NetCfg(meta["num_blocks"], O.ops_config_lib[meta["config"]], meta["use_layernorm"], meta["activation"], fixed=True)
    Pm = O.Params(torch.float64)
    for k, s in meta["param_shapes"].items():
        Pm.get_or_create(k, s)
    Pm.frozen = True
    ref = O.supernet_forward(Pm, ocfg, torch.tensor(z["int_x"]).double(), torch.tensor(z["cat_x"]), meta["choice"])
    out = eng.forward(int_x, cat_x).cpu().double()
    err = float((out - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    assert err <= 1e-5 * scale, "smoke: logits differ from the oracle by %.3e" % err
    _, loss_ref, _, _ = O.train_step(Pm, {}, ocfg, meta["choice"], torch.tensor(z["int_x"]).double(), torch.tensor(z["cat_x"]),
                                     torch.tensor(z["y"]).double(), lr=0.16)
    loss = eng.train_step(int_x, cat_x, y, lr=0.16)
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - float(loss_ref)) <= 1e-5 * max(1.0, abs(float(loss_ref)))
    after = eng.state_dict()
    worst = 0.0
    for k in ("_final.weight", "_blocks.6._nodes.0._linear.weight", "_embedding.0.weight", "_blocks.0._nodes.4._mha.in_proj_weight"):
        worst = max(worst, float((after[k].double() - Pm[k]).abs().max()))
    assert worst <= 1e-4, "smoke: parameters after one step differ from the oracle by %.3e" % worst
    print("smoke ok: logit err %.2e, loss %.6f (oracle %.6f), param err after step %.2e" % (err, float(loss.item()), float(loss_ref), worst))
