"""Row-sharded embedding tables on the GPU (nasrec_amd/sharded_tables.py): ShardedTableStep with the HIP engine behind its protocol
(EngineShardedOps: the network in `host_embedding` mode, the gather / dedup / row-Adagrad kernels on the rank's shards) against the
plain engine step with whole tables on a twin engine — single rank (the routing degenerates to the identity, every kernel and the
step composition run); the N > 1 routing is covered by the gloo world-2 test on the CPU."""
import os

import pytest
import torch

from helpers import GOLDEN, load_golden
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
    for _ in range(3):
        la = a.train_step(int_x, cat_x, y, 0.02, choice=choice)
        lb = step.step(int_x, cat_x, y, 0.02, choice=choice)
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
