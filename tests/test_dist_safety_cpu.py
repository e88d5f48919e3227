"""world_size-2 gloo tests (CPU) of what keeps N data-parallel replicas ONE model in the training CLIs (nasrec_amd/utils/dist.py,
train_utils._agreed_batches): identical seeding, rank-0 broadcast + checksum assertion, the step agreement over shards of
different length (no rank enters a collective the others skip), gradient averaging on the torch route, collective NaN exit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(WORLD),
                      NASREC_DIST_BACKEND="gloo", NASREC_PATH_SEED="5")
    from nasrec_amd.utils import dist as D
    from nasrec_amd.utils.train_utils import _agreed_batches
    import numpy as np
    r, w = D.init_from_env()
    assert (r, w) == (rank, WORLD)
    # identical seeding: the constructor inits draw the same numbers on every rank, the path stream too
    lin = torch.nn.Linear(5, 3)
    draws = [float(np.random.random()), float(torch.rand(1))]
    # a replica that went astray (rank 1 perturbs its weights): the checksum assertion must fire, the broadcast must repair it
    model = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.LayerNorm(4))
    if rank == 1:
        with torch.no_grad():
            model[0].weight.add_(1.0)
    caught = False
    try:
        D.assert_replicas_identical(model)
    except RuntimeError:
        caught = True
    D.broadcast_replica_state(model)
    D.assert_replicas_identical(model)
    # shards of different length: rank 0 holds 5 full batches, rank 1 holds 3 full batches + a short one; batch size 4
    B = 4
    n_full = 5 if rank == 0 else 3
    loader = [(torch.zeros(B, 2), torch.zeros(B, 1, dtype=torch.int64), torch.zeros(B)) for _ in range(n_full)]
    if rank == 1:
        loader.append((torch.zeros(2, 2), torch.zeros(2, 1, dtype=torch.int64), torch.zeros(2)))
    taken = 0
    for int_x, cat_x, y in _agreed_batches(loader, B, WORLD, None):
        assert len(y) == B
        t = torch.ones(1)
        dist.all_reduce(t)  # the step's collective: would hang if the ranks disagreed on the number of steps
        taken += 1
    # torch-route gradient averaging
    x = torch.full((3, 4), float(rank + 1))
    model.zero_grad()
    model(x).sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    D.allreduce_grads(model)
    avg = [p.grad.clone() for p in model.parameters()]
    nan_any = D.any_rank(rank == 1)
    nan_none = D.any_rank(False)
    out[rank] = dict(w=lin.weight.detach().clone(), draws=draws, caught=caught, taken=taken, local=local, avg=avg, nan_any=nan_any,
                     nan_none=nan_none, params=[p.detach().clone() for p in model.parameters()])
    dist.destroy_process_group()


def test_replicas_stay_one_model_under_gloo_world_2():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
    r0, r1 = out[0], out[1]
    assert torch.equal(r0["w"], r1["w"]) and r0["draws"] == r1["draws"], "ranks are not seeded identically"
    assert r0["caught"] and r1["caught"], "a perturbed replica passed the checksum assertion"
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    assert r0["taken"] == r1["taken"] == 3, "the epoch must end for every rank at the first step one rank cannot take"
    for l0, l1, a0, a1 in zip(r0["local"], r1["local"], r0["avg"], r1["avg"]):
        assert torch.allclose(a0, (l0 + l1) / 2) and torch.equal(a0, a1)
    assert r0["nan_any"] and r1["nan_any"] and not r0["nan_none"] and not r1["nan_none"]


def test_single_process_paths_are_untouched():
    from nasrec_amd.utils import dist as D
    from nasrec_amd.utils.train_utils import _agreed_batches
    assert D.world_info() == (0, 1)
    m = torch.nn.Linear(2, 2)
    D.broadcast_replica_state(m)
    D.assert_replicas_identical(m)
    D.allreduce_grads(m)
    assert D.any_rank(True) and not D.any_rank(False)
    loader = [(torch.zeros(4, 2), torch.zeros(4, 1), torch.zeros(4)), (torch.zeros(1, 2), torch.zeros(1, 1), torch.zeros(1))]
    assert len(list(_agreed_batches(loader, 4, 1, None))) == 2  # the short last batch still reaches the loop (evaluated, not trained on)
    a = D.StepAgreement()
    a.post(True)
    assert a.take() is True
