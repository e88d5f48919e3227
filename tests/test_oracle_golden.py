"""Pins oracle/nasrec_oracle.py to vectors produced by the real reference (tests/golden/make_golden.py).
CPU-only; the reference itself is not needed."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, check_trajectory, golden_files, load_golden, oracle_cfg, oracle_params, proj_checksum
from oracle import nasrec_oracle as O

NPZ = golden_files("fixed_*.npz") + golden_files("supernet_*.npz")


@pytest.mark.parametrize("path", NPZ, ids=[os.path.basename(p)[:-4] for p in NPZ])
def test_forward_fp64_and_fp32(path):
    z, meta = load_golden(path)
    cfg = oracle_cfg(meta)
    for dtype, key, tol in ((torch.float64, "logits_f64", 1e-9), (torch.float32, "logits_f32", 2e-5)):
        P = oracle_params(meta, dtype)
        out = O.supernet_forward(P, cfg, torch.tensor(z["int_x"]).to(dtype), torch.tensor(z["cat_x"]), meta["choice"])
        ref = z[key]
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(out.numpy() - ref).max() <= tol * scale, (key, np.abs(out.numpy() - ref).max())
        loss = O.bce_with_logits_mean(out, torch.tensor(z["y"]).to(dtype))
        assert abs(float(loss) - float(z["loss_" + key[-3:]])) <= tol * max(1.0, abs(float(loss)))


@pytest.mark.parametrize("path", NPZ, ids=[os.path.basename(p)[:-4] for p in NPZ])
def test_lazy_param_creation_matches_reference_state_dict(path):
    """Lazy creation in the oracle (full-path warm-up in supernet mode) yields exactly the reference's
    state_dict keys and shapes, i.e. the same deleted-projection rules."""
    z, meta = load_golden(path)
    cfg = oracle_cfg(meta)
    P = O.Params(torch.float64)
    warm = meta["choice"] if cfg.fixed else O.full_path_choice(cfg)
    O.supernet_forward(P, cfg, torch.tensor(z["int_x"]).double(), torch.tensor(z["cat_x"]), warm, num_embeddings=meta["tables"])
    got = {k: list(v.shape) for k, v in P.items()}
    assert got == meta["param_shapes"]


@pytest.mark.parametrize("path", NPZ, ids=[os.path.basename(p)[:-4] for p in NPZ])
def test_grads_and_three_adagrad_steps(path):
    z, meta = load_golden(path)
    cfg = oracle_cfg(meta)
    P = oracle_params(meta, torch.float64)
    int_x, cat_x, y = torch.tensor(z["int_x"]).double(), torch.tensor(z["cat_x"]), torch.tensor(z["y"]).double()
    state = {}
    losses, norms = [], []
    for step in range(meta["n_steps"]):
        before = {k: v.clone() for k, v in P.items()} if step == 0 else None
        _, loss, total, gd = O.train_step(P, state, cfg, meta["choice"], int_x, cat_x, y, lr=meta["lr"], clip=5.0, eps=1e-2)
        losses.append(float(loss))
        norms.append(float(total))
        if step == 0:
            # un-clipped gradient checksums: undo the clip factor
            coef = min(1.0, 5.0 / (float(total) + 1e-6))
            assert sorted(set(P.keys()) - set(gd.keys())) == sorted(meta["grad_none"])
            for k, (dot, nrm) in meta["grads"].items():
                d, n = proj_checksum(k, gd[k] / coef)
                assert abs(n - nrm) <= 1e-9 * max(1.0, nrm), k
                assert abs(d - dot) <= 1e-9 * max(1.0, nrm), k
    check_trajectory(z, meta, losses, norms, P, rel_params=1e-9, rel_delta=1e-8, rel_loss=1e-9)
    out = O.supernet_forward(P, cfg, int_x, cat_x, meta["choice"])
    assert np.abs(out.numpy() - z["logits_after_f64"]).max() <= 1e-9 * max(1.0, np.abs(z["logits_after_f64"]).max())


def test_block_outputs_cfg1():
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_criteo_xlarge.npz"))
    cfg = oracle_cfg(meta)
    P = oracle_params(meta, torch.float64)
    rec = {}
    O.supernet_forward(P, cfg, torch.tensor(z["int_x"]).double(), torch.tensor(z["cat_x"]), meta["choice"], record=rec)
    for i in range(meta["num_blocks"]):
        assert np.abs(rec["dense"][i].numpy() - z["block%d_dense" % i]).max() < 1e-9
        assert np.abs(rec["sparse"][i].numpy() - z["block%d_sparse" % i]).max() < 1e-9


def test_sampler_traces():
    traces = json.load(open(os.path.join(GOLDEN, "samplers.json")))
    assert len(traces) >= 30
    for tr in traces:
        cfg = O.NetCfg(tr["num_blocks"], O.ops_config_lib[tr["space"]], True)
        s = O.PathSampler(cfg, tr["strategy"], tr["anypath_choice"], tr["supernet_training_steps"], tr.get("candidate_choices"))
        # one full-path warm-up forward advanced every counter by one (train_utils.py:431-432)
        s.net_counter += tr["warmup_forwards"]
        s.block_counter = [c + tr["warmup_forwards"] for c in s.block_counter]
        np.random.seed(tr["seed"])
        for want in tr["choices"]:
            got = s.sample()
            assert json.loads(json.dumps(got)) == want, (tr["strategy"], tr["anypath_choice"], tr["supernet_training_steps"])


def test_lr_traces():
    lr = json.load(open(os.path.join(GOLDEN, "lr.json")))
    c = lr["cosine_cfg1"]
    n = c["first_cycle_steps"]
    seq = O.cosine_warmup_restarts_lrs(n, c["first_cycle_steps"], c["max_lr"], c["min_lr"], c["warmup_steps"])
    for t, v in c["lrs"].items():
        assert abs(seq[int(t)] - v) <= 1e-15 + 1e-12 * abs(v), t
    assert seq[0] == 1e-8  # first optimizer step runs at min_lr (lr_schedule.py:91-95)
    c = lr["cosine_restarts"]
    seq = O.cosine_warmup_restarts_lrs(len(c["lrs"]), c["first_cycle_steps"], c["max_lr"], c["min_lr"], c["warmup_steps"],
                                       c["cycle_mult"], c["gamma"])
    assert np.allclose(seq, c["lrs"], rtol=1e-12, atol=0)
    c = lr["constant_warmup"]
    assert np.allclose(O.constant_with_warmup_lrs(len(c["lrs"]), c["base_lr"], c["num_warmup_steps"]), c["lrs"], rtol=1e-12, atol=0)


def test_tril_order_matches_torch():
    for n in (2, 9, 40, 46):
        li, lj = O.tril_pairs(n)
        t = torch.tril_indices(n, n, offset=-1)
        assert li == t[0].tolist() and lj == t[1].tolist()
