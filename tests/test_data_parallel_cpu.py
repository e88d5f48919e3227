"""world_size-2 gloo test (CPU) of the data-parallel exchange used for N > 1 GPUs (nasrec_amd/parallel.py):
after the exchange every rank holds (a) the SUM of the flat dense gradients, (b) the ids and per-sample embedding-row
gradients of the GLOBAL batch in rank order; a row-sparse clip + Adagrad over (b) then equals the dense reference update
(clip_grad_norm_ + torch.optim.Adagrad) of a single process at the global batch — on every rank identically."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nasrec_amd.parallel import all_gather_rows_async, exchange_gradients

WORLD, B, FS, ROWS = 2, 6, 3, [4, 50, 7]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rowsparse_update(tables, state, idx, g, coef, lr, eps):
    """what emb_dedup + adagrad_rows compute: first occurrence leads, duplicates summed, touched rows updated"""
    for f in range(idx.shape[1]):
        seen = {}
        for b in range(idx.shape[0]):
            seen.setdefault(int(idx[b, f]), []).append(b)
        for row, bs in seen.items():
            gs = sum(g[b, f] for b in bs) * coef
            state[f][row] += gs * gs
            tables[f][row] -= lr * gs / (state[f][row].sqrt() + eps)


def _worker(rank, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    g = torch.Generator().manual_seed(100 + rank)
    flat_g = torch.randn(37, generator=g, dtype=torch.float64)
    cat = torch.stack([torch.randint(0, n, (B,), generator=g) for n in ROWS], 1)
    sg = torch.randn(B, FS, 16, generator=g, dtype=torch.float64) * 0.1
    cat_all = torch.zeros(WORLD * B, FS, dtype=torch.int64)
    sg_all = torch.zeros(WORLD * B * FS * 16, dtype=torch.float64)
    local_flat = flat_g.clone()
    exchange_gradients(flat_g, cat, sg, cat_all, sg_all)
    # the overlapped form DataParallelStep.step uses: asynchronous collectives, waited for in front of the optimizer
    cat_all2, sg_all2, flat2 = torch.zeros_like(cat_all), torch.zeros_like(sg_all), local_flat.clone()
    pending = [all_gather_rows_async(cat_all2, cat), all_gather_rows_async(sg_all2, sg),
               dist.all_reduce(flat2, op=dist.ReduceOp.SUM, async_op=True)]
    for w in pending:
        if w is not None:
            w.wait()
    assert torch.equal(cat_all2, cat_all) and torch.equal(sg_all2, sg_all) and torch.equal(flat2, flat_g)
    out[rank] = dict(local_flat=local_flat, flat=flat_g, cat=cat, sg=sg, cat_all=cat_all, sg_all=sg_all.view(WORLD * B, FS, 16))
    dist.destroy_process_group()


def test_exchange_and_rowsparse_equivalence():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
    r0, r1 = out[0], out[1]
    # (a) all-reduce: both ranks hold the same sum
    assert torch.equal(r0["flat"], r1["flat"])
    assert torch.allclose(r0["flat"], r0["local_flat"] + r1["local_flat"])
    # (b) all-gather in rank order, identical everywhere
    for r in (r0, r1):
        assert torch.equal(r["cat_all"], torch.cat([r0["cat"], r1["cat"]], 0))
        assert torch.equal(r["sg_all"], torch.cat([r0["sg"], r1["sg"]], 0))
    # (c) row-sparse update over the gathered rows == dense single-process reference at the global batch
    idx, g = r0["cat_all"], r0["sg_all"]
    tables = [torch.nn.Parameter(torch.randn(n, 16, dtype=torch.float64, generator=torch.Generator().manual_seed(5 + i)))
              for i, n in enumerate(ROWS)]
    dense = torch.nn.Parameter(torch.randn(37, dtype=torch.float64, generator=torch.Generator().manual_seed(9)))
    mine_t = [t.detach().clone() for t in tables]
    mine_s = [torch.zeros_like(t) for t in mine_t]
    mine_d, mine_ds = dense.detach().clone(), torch.zeros(37, dtype=torch.float64)
    opt = torch.optim.Adagrad(tables + [dense], lr=0.16, eps=1e-2)
    for step in range(2):
        for f in range(FS):
            tables[f].grad = torch.zeros_like(tables[f]).index_add_(0, idx[:, f], g[:, f])
        dense.grad = r0["flat"].clone()
        total = torch.nn.utils.clip_grad_norm_(tables + [dense], 0.3)
        opt.step()
        # engine-side arithmetic
        ss = float((r0["flat"] ** 2).sum())
        for f in range(FS):
            seen = {}
            for b in range(idx.shape[0]):
                seen.setdefault(int(idx[b, f]), []).append(b)
            for bs in seen.values():
                ss += float((sum(g[b, f] for b in bs) ** 2).sum())
        assert abs(ss ** 0.5 - float(total)) < 1e-12
        coef = min(1.0, 0.3 / (ss ** 0.5 + 1e-6))
        _rowsparse_update(mine_t, mine_s, idx, g, coef, 0.16, 1e-2)
        gd = r0["flat"] * coef
        mine_ds += gd * gd
        mine_d -= 0.16 * gd / (mine_ds.sqrt() + 1e-2)
    for f in range(FS):
        assert torch.allclose(mine_t[f], tables[f].data, rtol=0, atol=1e-12)
    assert torch.allclose(mine_d, dense.data, rtol=0, atol=1e-12)


# ------------------------------------------------------------------------------------------------------------------
# the whole data-parallel step against a single process at the global batch, with the oracle as the per-rank engine
# ------------------------------------------------------------------------------------------------------------------
def _dp_oracle_worker(rank, port, out):
    """What DataParallelStep.step does on each rank, with the fp64 oracle standing in for the HIP engine: local
    forward/backward with d loss / d logits pre-scaled by 1 / (B_local * world), asynchronous all-gather of (ids, per-sample
    embedding-row gradients) and all-reduce of the dense gradients, then the same global-batch clip + row-sparse Adagrad on
    every rank."""
    import json
    from helpers import GOLDEN, load_golden, oracle_cfg, oracle_params
    from oracle import nasrec_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_criteo_autoctr.npz"))
    cfg, P = oracle_cfg(meta), oracle_params(meta)
    int_x, cat_x, y = torch.tensor(z["int_x"]).double(), torch.tensor(z["cat_x"]), torch.tensor(z["y"]).double().view(-1, 1)
    Bg = int_x.shape[0]
    Bl = Bg // WORLD
    sl = slice(rank * Bl, (rank + 1) * Bl)
    Fs = cat_x.shape[1]
    lr, eps, clip = 0.05, 1e-2, 5.0
    state = {}
    for step in range(2):
        leaves = {k: v.detach().requires_grad_(True) for k, v in P.items() if not k.startswith("_embedding.")}
        Pl = O.Params(P.dtype, frozen=True)
        Pl.update(leaves)
        rows = torch.stack([P["_embedding.%d.weight" % f][cat_x[sl, f]] for f in range(Fs)], 1).detach().requires_grad_(True)  # [Bl,Fs,16]
        for f in range(Fs):  # the stem as a gather of leaf rows, so that d loss / d rows is the per-sample row gradient
            Pl["_embedding.%d.weight" % f] = rows[:, f]
        ids_local = torch.arange(Bl).view(Bl, 1).expand(Bl, Fs).contiguous()
        logits = O.supernet_forward(Pl, cfg, int_x[sl], ids_local, meta["choice"])
        per_sample = torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), y[sl].view(-1), reduction="sum")
        loss = per_sample / (Bl * WORLD)  # == grad_scale 1/(B_local*world) folded into d logits
        names = list(leaves)
        grads = torch.autograd.grad(loss, [leaves[k] for k in names] + [rows], allow_unused=True)
        flat = torch.cat([(g if g is not None else torch.zeros_like(leaves[k])).reshape(-1) for k, g in zip(names, grads[:-1])])
        sg = grads[-1].contiguous()
        cat_all, sg_all = torch.zeros(Bg, Fs, dtype=torch.int64), torch.zeros(Bg * Fs * 16, dtype=torch.float64)
        pending = [all_gather_rows_async(cat_all, cat_x[sl].contiguous()), all_gather_rows_async(sg_all, sg),
                   dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)]
        for w in pending:
            if w is not None:
                w.wait()
        sg_all = sg_all.view(Bg, Fs, 16)
        # global-batch optimizer, identical on every rank: dedup rows (ascending sample order), clip over dense + unique rows
        uniq = []
        for f in range(Fs):
            ids, inv = torch.unique(cat_all[:, f], return_inverse=True)
            uniq.append((ids, torch.zeros(len(ids), 16, dtype=torch.float64).index_add_(0, inv, sg_all[:, f])))
        total = torch.sqrt(flat.pow(2).sum() + sum(g.pow(2).sum() for _, g in uniq))
        coef = min(1.0, clip / (float(total) + 1e-6))
        with torch.no_grad():
            off = 0
            for k in names:
                n = leaves[k].numel()
                g = flat[off:off + n].view_as(P[k]) * coef
                off += n
                if not bool((g != 0).any()) and k not in state and k in meta["grad_none"]:
                    continue  # parameters outside the path: torch skips grad=None
                st = state.setdefault(k, torch.zeros_like(P[k]))
                O.adagrad_step_(P[k], g, st, lr, eps)
            for f, (ids, g) in enumerate(uniq):
                k = "_embedding.%d.weight" % f
                st = state.setdefault(k, torch.zeros_like(P[k]))
                g = g * coef
                st[ids] += g * g
                P[k][ids] -= lr * g / (st[ids].sqrt() + eps)
    out[rank] = {k: v.detach().clone() for k, v in P.items()}
    dist.destroy_process_group()


def test_data_parallel_step_equals_single_process_at_the_global_batch():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import GOLDEN, load_golden, oracle_cfg, oracle_params
    from oracle import nasrec_oracle as O
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_oracle_worker, args=(port, out), nprocs=WORLD, join=True)
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_criteo_autoctr.npz"))
    cfg, P = oracle_cfg(meta), oracle_params(meta)
    int_x, cat_x, y = torch.tensor(z["int_x"]).double(), torch.tensor(z["cat_x"]), torch.tensor(z["y"]).double().view(-1, 1)
    state = {}
    for step in range(2):
        O.train_step(P, state, cfg, meta["choice"], int_x, cat_x, y, lr=0.05)
    for r in range(WORLD):
        for k, v in P.items():
            assert torch.allclose(out[r][k], v, rtol=0, atol=1e-10), (r, k)
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k  # replicas stay bit-identical
