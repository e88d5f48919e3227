"""Row-sharded embedding tables on the GPU (nasrec_amd/sharded_tables.py): ShardedTableStep with the HIP engine behind its protocol
(EngineShardedOps: the network in `host_embedding` mode, the gather / dedup / row-Adagrad kernels on the rank's shards) against the
REFERENCE's golden trajectories (tests/golden/*.npz: three clipped-Adagrad steps of the real reference in fp64), against the fp64 oracle
on alternating sampled paths, and against the plain engine step with whole tables — single rank (the routing degenerates to the
identity, every kernel and the step composition run); the N > 1 routing is covered by the gloo world-2 test on the CPU."""
import os

import pytest
import torch

from helpers import GOLDEN, check_trajectory, load_golden, oracle_cfg, oracle_params
from nasrec_amd import plan as P
from nasrec_amd.engine import SupernetEngine
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.sharded_tables import EngineShardedOps, RowShardedTables, ShardedTableStep
from oracle import nasrec_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,B", [("fixed_criteo_autoctr", 96), ("supernet_autoctr_single", 1200)])
def test_sharded_step_matches_the_whole_table_step(case, B):
    z, meta = load_golden(os.path.join(GOLDEN, case + ".npz"))
    fixed = meta["mode"] == "fixed"
    cfg = P.NetConfig(meta["num_blocks"], ops_config_lib[meta["config"]], meta["use_layernorm"], meta["activation"], fixed=fixed)
    Fd, Fs, tables = z["int_x"].shape[1], z["cat_x"].shape[1], meta["tables"]
    int_x, cat_x, y = O.synthetic_batch(B, Fd, tables, seed=9)
    cat_x[:, 0] = cat_x[0, 0]  # one row hit by every sample
    int_x, cat_x, y = int_x.cuda(), cat_x.cuda(), y.view(-1).cuda()
    choice = meta["choice"]
    a = SupernetEngine(cfg, Fd, Fs, tables, warm_choice=choice if fixed else None)
    a.init_weights(seed=4)
    b = SupernetEngine(cfg, Fd, Fs, tables, warm_choice=choice if fixed else None, host_embedding=True)
    b.load_params({k: v for k, v in a.state_dict().items() if not k.startswith("_embedding.")})
    t = RowShardedTables(tables, "cuda", init_fn=lambda f, lo, hi: a.tables[f][lo:hi].clone())
    step = ShardedTableStep(EngineShardedOps(b, clip=5.0, eps=1e-2), t, B, clip=5.0, eps=1e-2)
    # (the B = 1200 supernet path DIVERGES at this learning rate — loss 0.55, 1.26, 16.7 — which makes the comparison a sharp one: one ulp of
    # difference in the clip coefficient is 1e-4 of gradient a step later.  ShardedTableStep computes the coefficient in the fp32
    # arithmetic of the whole-table step's kernel for this reason.)
    lr = 0.02
    for _ in range(3):
        la = a.train_step(int_x, cat_x, y, lr, choice=choice)
        lb = step.step(int_x, cat_x, y, lr, choice=choice)
        torch.cuda.synchronize()
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(la)))
        assert abs(float(step.last_norm) - float(a.clip_out[1])) <= 1e-5 * max(1.0, float(a.clip_out[1]))
    a.check_indices()
    b.check_indices()
    scale = float(a.flat_p.abs().max())
    assert float((a.flat_p - b.flat_p).abs().max()) <= 2e-6 * scale, "dense parameters"
    for f in range(Fs):
        assert torch.allclose(t.whole_table(f), a.tables[f], rtol=0, atol=2e-6), "table %d" % f
        touched = torch.zeros(tables[f], dtype=torch.bool, device="cuda")
        touched[cat_x[:, f]] = True
        assert torch.equal(t.state[f][:tables[f]][~touched], torch.zeros_like(t.state[f][:tables[f]][~touched]))


def _sharded_engine(z, meta):
    """host_embedding engine + row-sharded tables holding the fixture's name-seeded parameters"""
    fixed = meta["mode"] == "fixed"
    cfg = P.NetConfig(meta["num_blocks"], ops_config_lib[meta["config"]], meta["use_layernorm"], meta["activation"], fixed=fixed,
                      last_n_blocks_out=meta.get("last_n_blocks_out", 1))
    Fd, Fs, tables = z["int_x"].shape[1], z["cat_x"].shape[1], meta["tables"]
    eng = SupernetEngine(cfg, Fd, Fs, tables, warm_choice=meta["choice"] if fixed else None, host_embedding=True)
    missing = eng.load_params({k: O.seeded_param(k, shp) for k, shp in meta["param_shapes"].items() if not k.startswith("_embedding.")})
    assert missing == []
    t = RowShardedTables(tables, "cuda", init_fn=lambda f, lo, hi: torch.tensor(
        O.seeded_param("_embedding.%d.weight" % f, meta["param_shapes"]["_embedding.%d.weight" % f]), dtype=torch.float32)[lo:hi])
    return eng, t


@pytest.mark.parametrize("case", ["fixed_criteo_autoctr", "fixed_kdd_autoctr", "fixed_criteo_xlarge", "supernet_autoctr_single", "supernet_xlarge_any"])
def test_sharded_step_follows_the_reference_trajectory(case):
    """the golden fixture's three clipped-Adagrad steps (the real reference, fp64) through ShardedTableStep: per-step loss and
    pre-clip gradient norm, every parameter (dense and every table, via all-gather of the shards) after the last step — the same
    check, with the same tolerances, as the plain engine step gets in tests/test_parity_gpu.py::test_three_adagrad_steps"""
    from test_parity_gpu import TRAJ_REL_DELTA, TRAJ_REL_LOSS, TRAJ_REL_PARAMS
    z, meta = load_golden(os.path.join(GOLDEN, case + ".npz"))
    eng, t = _sharded_engine(z, meta)
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda().view(-1)
    step = ShardedTableStep(EngineShardedOps(eng, clip=5.0, eps=1e-2), t, int_x.shape[0], clip=5.0, eps=1e-2)
    losses, norms = [], []
    for _ in range(meta["n_steps"]):
        loss = step.step(int_x, cat_x, y, meta["lr"], choice=meta["choice"])
        torch.cuda.synchronize()
        losses.append(float(loss.item()))
        norms.append(float(step.last_norm))
    eng.check_indices()
    params = eng.state_dict()
    for f in range(len(meta["tables"])):
        params["_embedding.%d.weight" % f] = t.whole_table(f).cpu()
    check_trajectory(z, meta, losses, norms, params, rel_params=TRAJ_REL_PARAMS, rel_delta=TRAJ_REL_DELTA, rel_loss=TRAJ_REL_LOSS)


def test_sharded_step_on_alternating_paths_matches_the_oracle():
    """Weight-sharing supernet, a DIFFERENT path every step (the three choices of the supernet_xlarge_* fixtures in turn, twice): a
    path writes only its own ranges of the gradient arena, the others keep an earlier path's gradients — the clip norm and the update
    must cover the current path only (torch skips grad None).  Against the fp64 oracle's train_step on the same batch."""
    zs = [load_golden(os.path.join(GOLDEN, "supernet_xlarge_%s.npz" % n)) for n in ("any", "single", "full")]
    z, meta = zs[0]
    assert all(m["param_shapes"] == meta["param_shapes"] and m["tables"] == meta["tables"] for _, m in zs)
    choices = [m["choice"] for _, m in zs]
    assert len({str(c) for c in choices}) == 3
    eng, t = _sharded_engine(z, meta)
    int_x, cat_x, y = torch.tensor(z["int_x"]), torch.tensor(z["cat_x"]), torch.tensor(z["y"])
    step = ShardedTableStep(EngineShardedOps(eng, clip=5.0, eps=1e-2), t, int_x.shape[0], clip=5.0, eps=1e-2)
    cfg, Pm = oracle_cfg(meta), oracle_params(meta)
    state = {}
    lr = 0.01
    for k in range(6):
        ch = choices[k % 3]
        ref = O.train_step(Pm, state, cfg, ch, int_x.double(), cat_x, y.double().view(-1, 1), lr=lr)
        loss = step.step(int_x.cuda(), cat_x.cuda(), y.cuda().view(-1), lr, choice=ch)
        torch.cuda.synchronize()
        _, ref_loss, ref_norm, _ = ref
        assert abs(float(loss) - float(ref_loss)) <= 1e-5 * max(1.0, abs(float(ref_loss))), (k, float(loss), float(ref_loss))
        assert abs(float(step.last_norm) - float(ref_norm)) <= 1e-4 * max(1.0, float(ref_norm)), (k, float(step.last_norm), float(ref_norm))
    sd = eng.state_dict()
    for k, v in Pm.items():
        got = t.whole_table(int(k.split(".")[1])).cpu() if k.startswith("_embedding.") else sd[k]
        tol = 1e-4 * max(1.0, float(v.abs().max()))
        assert float((got.double().reshape(v.shape) - v).abs().max()) <= tol, k
