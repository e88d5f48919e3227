"""The data-parallel step of the HIP engine at WORLD SIZE 2 on one GPU: two rank processes share cuda:0 and exchange over gloo (RCCL
refuses two ranks on one device; gloo moves device tensors through the host) — the engine-side code of an N-GPU step with N > 1 for
real: half batches, row gradients written into the all-gather's send buffer, the last pieces' dense gradients riding behind them
and summed in rank order, the all-reduced pieces in front, the global-batch row dedup over the receive buffer's rank layout
(one-launch kernels at 2 x 4 samples; the two-halves form with its id half when NASREC_DEDUP_SPLIT_MAX_B admits the global batch),
clip + Adagrad over the gathered batch on both ranks.  Expected: the plain engine step of ONE process at the global batch (reference
semantics at that batch size: nasrec/utils/train_utils.py:262-286), and bit-identical replicas.  (Eager exchange: gloo collectives
cannot be captured into a graph; the captured form runs in tests/test_parity_gpu.py on a single-rank RCCL group.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, case, steps, pack, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from helpers import GOLDEN, load_golden
    from nasrec_amd import parallel
    from nasrec_amd.parallel import DataParallelStep
    from test_parity_gpu import build_engine
    parallel.PACK_TAIL_FLOATS = pack
    z, meta = load_golden(os.path.join(GOLDEN, case + ".npz"))
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda().view(-1)
    Bl = int_x.shape[0] // WORLD
    sl = slice(rank * Bl, (rank + 1) * Bl)
    eng = build_engine(z, meta)
    fixed = meta["mode"] == "fixed"
    dp = DataParallelStep(eng, meta["choice"] if fixed else None, Bl, clip=5.0, eps=1e-2, graph=False)
    assert dp.exchange and dp.world == WORLD
    losses = []
    for _ in range(steps):
        loss = dp.step(int_x[sl].contiguous(), cat_x[sl].contiguous(), y[sl].contiguous(), meta["lr"], choice=meta["choice"])
        torch.cuda.synchronize()
        losses.append(float(loss))
    eng.check_indices()
    out[rank] = dict(params={k: v.cpu() for k, v in eng.state_dict().items()}, losses=losses, tail=dp.tail_n, ids_half=dp.ids_half is not None)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,pack", [("fixed_criteo_xlarge", 65536), ("fixed_criteo_xlarge", 0), ("fixed_kdd_autoctr", 65536), ("supernet_xlarge_any", 0)])
def test_two_ranks_on_one_gpu_equal_one_process_at_the_global_batch(case, pack):
    from helpers import GOLDEN, load_golden
    from test_parity_gpu import build_engine
    z, meta = load_golden(os.path.join(GOLDEN, case + ".npz"))
    steps = 3
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(port, case, steps, pack, out), nprocs=WORLD, join=True)
    # one process, whole batch
    int_x, cat_x, y = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda().view(-1)
    eng = build_engine(z, meta)
    ref_losses = []
    for _ in range(steps):
        ref_losses.append(float(eng.train_step(int_x, cat_x, y, meta["lr"], choice=meta["choice"])))
        torch.cuda.synchronize()
    ref = {k: v.cpu() for k, v in eng.state_dict().items()}
    r0, r1 = out[0], out[1]
    if pack == 0:
        assert r0["tail"] == 0
    elif case == "fixed_criteo_xlarge":  # (a network whose backward is one piece has no tail to pack)
        assert r0["tail"] > 0
    for k in ref:
        assert torch.equal(r0["params"][k], r1["params"][k]), "replicas differ: %s" % k
        scale = max(1.0, float(ref[k].abs().max()))
        assert float((r0["params"][k] - ref[k]).abs().max()) <= 2e-5 * scale, k
    # each rank's loss is its share of the global mean (the 1 / (B world) factor is folded into d loss / d logits; the reported loss is
    # the rank's own mean): their average is the global mean
    for t in range(steps):
        assert abs(0.5 * (r0["losses"][t] + r1["losses"][t]) - ref_losses[t]) <= 1e-4 * max(1.0, abs(ref_losses[t])), (t, r0["losses"][t], r1["losses"][t], ref_losses[t])
