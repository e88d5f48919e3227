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
# DataParallelStep.step ITSELF (nasrec_amd/parallel.py) under gloo, world 2: the same step code the GPUs run, driven through
# the engine protocol by a CPU stand-in whose compute is the fp64 oracle.  Checked against a single process at the global
# batch: fixed sub-network, weight-sharing supernet with a freshly sampled path every step (shared np.random seed), and a
# global batch > 8192.
# ------------------------------------------------------------------------------------------------------------------
class OracleEngine:
    """The data-parallel protocol (dp_plan / dp_optimizer / flat_g / Fs / device / cfg.fixed) on CPU tensors, oracle inside."""

    def __init__(self, cfg, P, Fs):
        from nasrec_amd.parallel import DPPlan, coalesce_ranges
        self._DPPlan, self._coalesce = DPPlan, coalesce_ranges
        self.cfg, self.P, self.Fs = cfg, P, Fs
        self.device = torch.device("cpu")
        self.grad_dtype = torch.float64
        self.names = [k for k in P if not k.startswith("_embedding.")]
        self.offsets, off = {}, 0
        for k in self.names:
            self.offsets[k] = off
            off += (P[k].numel() + 3) // 4 * 4  # 16-byte aligned arena slots, like the engine
        self.flat_numel = off
        self.flat_g = torch.zeros(off, dtype=torch.float64)
        self.state = {}
        self.sent_bytes = 0

    def dp_plan(self, choice, B, grad_scale, clip, eps, graph):
        from oracle import nasrec_oracle as O
        eng, P, cfg, Fs = self, self.P, self.cfg, self.Fs
        plan = self._DPPlan()
        plan.cat_local = torch.zeros(B, Fs, dtype=torch.int64)
        plan.sparse_grad = torch.zeros(B * Fs * 16, dtype=torch.float64)
        plan.loss = torch.zeros(1, dtype=torch.float64)
        box = {}

        def stage(int_x, cat_x, y, lr):
            box.update(int_x=int_x, y=y, lr=lr)
            plan.cat_local.copy_(cat_x)

        def fwd_bwd():
            leaves = {k: P[k].detach().requires_grad_(True) for k in eng.names}
            Pl = O.Params(P.dtype, frozen=True)
            Pl.update(leaves)
            rows = torch.stack([P["_embedding.%d.weight" % f][plan.cat_local[:, f]] for f in range(Fs)], 1).detach().requires_grad_(True)
            for f in range(Fs):  # the stem as a gather of leaf rows: d loss / d rows is the per-sample row gradient
                Pl["_embedding.%d.weight" % f] = rows[:, f]
            ids = torch.arange(B).view(B, 1).expand(B, Fs).contiguous()
            logits = O.supernet_forward(Pl, cfg, box["int_x"], ids, choice)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), box["y"].view(-1), reduction="sum") * grad_scale
            grads = torch.autograd.grad(loss, [leaves[k] for k in eng.names] + [rows], allow_unused=True)
            eng.flat_g.zero_()  # what the engine's memset does: parameters outside the path see g = 0
            used = []
            for k, g in zip(eng.names, grads[:-1]):
                if g is not None:
                    eng.flat_g[eng.offsets[k]:eng.offsets[k] + g.numel()] = g.reshape(-1)
                    used.append(k)
            plan.sparse_grad.copy_(grads[-1].reshape(-1))
            plan.loss.fill_(float(loss.detach()))
            box["used"] = used

        plan.stage, plan.forward = stage, (lambda: None)
        if cfg.fixed:
            plan.segments = [(fwd_bwd, [(0, eng.flat_numel)])]
        else:
            # one bucket per block in reverse order, holding only the arena ranges of the parameters this path trained there
            # (what EngineDP derives from the backward program's block marks)
            def ranges_of(blk):
                def f():
                    names = [k for k in box["used"] if k.startswith("_blocks.%d." % blk) or (blk == cfg.num_blocks - 1 and k.startswith("_final."))]
                    return eng._coalesce([(eng.offsets[k], P[k].numel()) for k in names])
                return f
            lazy = [_LazyRanges(ranges_of(b)) for b in reversed(range(cfg.num_blocks))]
            plan.segments = [(fwd_bwd, lazy[0])] + [((lambda: None), r) for r in lazy[1:]]
        return plan

    def dp_path_spans(self, choice, B, grad_scale, clip, eps):
        """arena ranges of the parameters the path trains: what autograd leaves a gradient for (the real engine reads its plan)"""
        from oracle import nasrec_oracle as O
        P = self.P
        leaves = {k: P[k].detach().requires_grad_(True) for k in self.names}
        Pl = O.Params(P.dtype, frozen=True)
        Pl.update(leaves)
        rows = torch.zeros(2, self.Fs, 16, dtype=P.dtype)
        for f in range(self.Fs):
            Pl["_embedding.%d.weight" % f] = rows[:, f]
        Fd = getattr(self, "Fd", 3)
        logits = O.supernet_forward(Pl, self.cfg, torch.zeros(2, Fd, dtype=P.dtype), torch.arange(2).view(2, 1).expand(2, self.Fs).contiguous(), choice)
        grads = torch.autograd.grad(logits.sum(), [leaves[k] for k in self.names], allow_unused=True)
        return [(self.offsets[k], P[k].numel()) for k, g in zip(self.names, grads) if g is not None]

    def dp_optimizer(self, Bg, cat_all, sg_all, clip, eps, graph, rank_layout=None):
        from oracle import nasrec_oracle as O
        eng, P, Fs = self, self.P, self.Fs

        def run(plan=None, spans=None):
            # (spans: the union of the ranks' parameter ranges in per-rank path mode — Adagrad with g = 0 is a no-op, so walking every
            # parameter is the same update; the real engine restricts norm and update to the ranges)
            if rank_layout is None:
                sg = sg_all.view(Bg, Fs, 16)
            else:  # the receive buffer of the all-gather: per rank [Bl, Fs, 16] rows, then the dense gradients that rode along
                Bl, stride = rank_layout
                sg = sg_all.view(Bg // Bl, stride)[:, :Bl * Fs * 16].reshape(Bg, Fs, 16)
            uniq = []
            for f in range(Fs):
                ids, inv = torch.unique(cat_all[:, f], return_inverse=True)
                uniq.append((ids, torch.zeros(len(ids), 16, dtype=torch.float64).index_add_(0, inv, sg[:, f])))
            total = torch.sqrt(eng.flat_g.pow(2).sum() + sum(g.pow(2).sum() for _, g in uniq))
            coef = min(1.0, clip / (float(total) + 1e-6))
            with torch.no_grad():
                for k in eng.names:
                    g = eng.flat_g[eng.offsets[k]:eng.offsets[k] + P[k].numel()].view_as(P[k]) * coef
                    st = eng.state.setdefault(k, torch.zeros_like(P[k]))
                    O.adagrad_step_(P[k], g, st, eng.lr, eps)  # g = 0 leaves state and parameter untouched (Adagrad no-op)
                for f, (ids, g) in enumerate(uniq):
                    k = "_embedding.%d.weight" % f
                    st = eng.state.setdefault(k, torch.zeros_like(P[k]))
                    g = g * coef
                    st[ids] += g * g
                    P[k][ids] -= eng.lr * g / (st[ids].sqrt() + eps)
        return run


class _LazyRanges:
    """iterable of arena ranges that is only known once the segment before it has run (the stand-in learns the path's parameter
    set from autograd; the real engine knows it at compile time)"""

    def __init__(self, fn):
        self.fn = fn

    def __iter__(self):
        return iter(self.fn())


def _dp_case(name):
    """-> (cfg, P, batches [(int_x, cat_x, y)], choices or None, lr) for the three scenarios"""
    from helpers import GOLDEN, load_golden, oracle_cfg, oracle_params
    from oracle import nasrec_oracle as O
    if name.startswith("fixed"):
        z, meta = load_golden(os.path.join(GOLDEN, "fixed_criteo_autoctr.npz"))
        cfg, P = oracle_cfg(meta), oracle_params(meta)
        b = (torch.tensor(z["int_x"]).double(), torch.tensor(z["cat_x"]), torch.tensor(z["y"]).double().view(-1, 1))
        return cfg, P, [b, b], [meta["choice"]] * 2, 0.05
    nb, space, Bg, tables, steps = (3, "autoctr", 8, [40, 7, 300, 5], 3) if name == "sampled" else (1, "autoctr", 8400, [4, 90, 50000], 1)
    cfg = O.NetCfg(nb, O.ops_config_lib[space], True, "relu", fixed=False)
    P = O.Params(torch.float64)
    batches = []
    for s in range(steps):
        int_x, cat_x, y = O.synthetic_batch(Bg, 3, tables, seed=40 + s)
        batches.append((int_x.double(), cat_x, y.double().view(-1, 1)))
    with torch.no_grad():
        O.supernet_forward(P, cfg, batches[0][0][:4], batches[0][1][:4], O.full_path_choice(cfg), num_embeddings=tables)  # lazy params, full path
    P.frozen = True
    np.random.seed(1234)  # every rank draws the same paths from its own (identically seeded) global stream
    sampler = O.PathSampler(cfg, "default", "binomial-0.5")
    choices = [json.loads(json.dumps(sampler.sample(), default=lambda o: o.tolist() if hasattr(o, "tolist") else o.item())) for _ in range(steps)]
    return cfg, P, batches, choices, 0.02


def _dp_step_worker(rank, port, name, out):
    from nasrec_amd import parallel
    from nasrec_amd.parallel import DataParallelStep
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    # "fixed": the dense gradients of the backward's last piece ride in the row-gradient all-gather (here: all of them — the stand-in
    # has one piece) and are added in rank order; "fixed_allreduce": every piece is all-reduced
    parallel.PACK_TAIL_FLOATS = {"fixed": 1 << 30, "fixed_allreduce": 0}.get(name, parallel.PACK_TAIL_FLOATS)
    cfg, P, batches, choices, lr = _dp_case(name)
    Fs = batches[0][1].shape[1]
    Bl = batches[0][0].shape[0] // WORLD
    eng = OracleEngine(cfg, P, Fs)
    dp = DataParallelStep(eng, choices[0] if cfg.fixed else None, Bl, clip=5.0, eps=1e-2, graph=False)
    assert dp.exchange and dp.world == WORLD
    if name.startswith("fixed"):
        assert (dp.tail_n > 0) == (name == "fixed"), dp.tail_n
    sl = slice(rank * Bl, (rank + 1) * Bl)
    losses = []
    for (int_x, cat_x, y), ch in zip(batches, choices):
        eng.lr = lr
        loss = dp.step(int_x[sl], cat_x[sl].contiguous(), y[sl], lr, choice=None if cfg.fixed else ch)
        losses.append(float(loss))
    out[rank] = ({k: v.detach().clone() for k, v in P.items()}, losses)
    dist.destroy_process_group()


import json  # noqa: E402

import pytest  # noqa: E402


@pytest.mark.parametrize("name", ["fixed", "fixed_allreduce", "sampled", "large_batch"])
def test_data_parallel_step_equals_single_process_at_the_global_batch(name):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import nasrec_oracle as O
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_step_worker, args=(port, name, out), nprocs=WORLD, join=True)
    cfg, P, batches, choices, lr = _dp_case(name)
    initial = {k: v.clone() for k, v in P.items()}
    state = {}
    touched = set()
    for (int_x, cat_x, y), ch in zip(batches, choices):
        _, _, _, gd = O.train_step(P, state, cfg, ch, int_x, cat_x, y, lr=lr)
        touched |= set(gd)
    for r in range(WORLD):
        for k, v in P.items():
            assert torch.allclose(out[r][0][k], v, rtol=0, atol=1e-10), (name, r, k)
    for k in out[0][0]:
        assert torch.equal(out[0][0][k], out[1][0][k]), k  # replicas stay bit-identical
    # the sum of the ranks' losses (each scaled by 1/(B_local*world)) is the global mean loss of the step
    if not name.startswith("fixed"):
        untouched = [k for k in P if k not in touched]
        assert untouched, "a sampled path leaves part of the supernet unused"
        for k in untouched:  # grad None in the reference: neither value nor Adagrad state may move
            assert torch.equal(out[0][0][k], initial[k]), k


# ------------------------------------------------------------------------------------------------------------------
# paths="per-rank" (SURVEY 8e optional mode; north star: "per-sample supernet paths shard data-parallel"): every rank trains its own
# sampled path; a parameter's gradient is the sum over the ranks whose path used it; clip + Adagrad over the union.  Checked against
# the fp64 oracle evaluating the two paths on the two half-batches.
# ------------------------------------------------------------------------------------------------------------------
def _per_rank_case():
    from oracle import nasrec_oracle as O
    tables, Bg, steps = [40, 7, 300, 5], 8, 3
    cfg = O.NetCfg(3, O.ops_config_lib["autoctr"], True, "relu", fixed=False)
    P = O.Params(torch.float64)
    batches = []
    for s in range(steps):
        int_x, cat_x, y = O.synthetic_batch(Bg, 3, tables, seed=90 + s)
        batches.append((int_x.double(), cat_x, y.double().view(-1, 1)))
    with torch.no_grad():
        O.supernet_forward(P, cfg, batches[0][0][:4], batches[0][1][:4], O.full_path_choice(cfg), num_embeddings=tables)
    P.frozen = True
    choices = []
    for r in range(WORLD):  # rank r draws from seed + r: different paths on the two ranks
        np.random.seed(4321 + r)
        sampler = O.PathSampler(cfg, "default", "binomial-0.5")
        choices.append([json.loads(json.dumps(sampler.sample(), default=lambda o: o.tolist() if hasattr(o, "tolist") else o.item())) for _ in range(steps)])
    return cfg, P, batches, choices, 0.02


def _per_rank_worker(rank, port, out):
    from nasrec_amd.parallel import DataParallelStep
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    cfg, P, batches, choices, lr = _per_rank_case()
    Fs = batches[0][1].shape[1]
    Bl = batches[0][0].shape[0] // WORLD
    eng = OracleEngine(cfg, P, Fs)
    dp = DataParallelStep(eng, None, Bl, clip=5.0, eps=1e-2, graph=False, paths="per-rank")
    sl = slice(rank * Bl, (rank + 1) * Bl)
    unions = []
    for s, (int_x, cat_x, y) in enumerate(batches):
        eng.lr = lr
        dp.step(int_x[sl], cat_x[sl].contiguous(), y[sl], lr, choice=choices[rank][s])
        unions.append(list(dp._last[1].union))
    out[rank] = ({k: v.detach().clone() for k, v in P.items()}, unions)
    dist.destroy_process_group()


def test_per_rank_paths_sum_gradients_over_the_ranks_that_used_a_parameter():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import nasrec_oracle as O
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_per_rank_worker, args=(port, out), nprocs=WORLD, join=True)
    cfg, P, batches, choices, lr = _per_rank_case()
    assert any(json.dumps(a) != json.dumps(b) for a, b in zip(choices[0], choices[1])), "the two ranks should train different paths"
    initial = {k: v.clone() for k, v in P.items()}
    state, touched = {}, set()
    names = list(P.keys())
    Bg = batches[0][0].shape[0]
    Bl = Bg // WORLD
    for s, (int_x, cat_x, y) in enumerate(batches):
        total = {}
        for r in range(WORLD):  # path r on half r; the loss is the mean over the GLOBAL batch
            sl = slice(r * Bl, (r + 1) * Bl)
            leaves = {k: v.detach().requires_grad_(True) for k, v in P.items()}
            Pl = O.Params(P.dtype, frozen=True)
            Pl.update(leaves)
            logits = O.supernet_forward(Pl, cfg, int_x[sl], cat_x[sl], choices[r][s])
            loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), y[sl].view(-1), reduction="sum") / Bg
            grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
            for k, g in zip(names, grads):
                if g is not None:
                    total[k] = total[k] + g if k in total else g.clone()
        norm = torch.sqrt(sum(g.pow(2).sum() for g in total.values()))
        coef = min(1.0, 5.0 / (float(norm) + 1e-6))
        with torch.no_grad():
            for k, g in total.items():
                st = state.setdefault(k, torch.zeros_like(P[k]))
                O.adagrad_step_(P[k], g * coef, st, lr, 1e-2)
        touched |= set(total)
    for r in range(WORLD):
        for k, v in P.items():
            assert torch.allclose(out[r][0][k], v, rtol=0, atol=1e-10), (r, k)
    for k in out[0][0]:
        assert torch.equal(out[0][0][k], out[1][0][k]), k  # replicas stay bit-identical
    assert out[0][1] == out[1][1], "every rank must reduce the same union of ranges"
    untouched = [k for k in P if k not in touched]
    assert untouched
    for k in untouched:
        assert torch.equal(out[0][0][k], initial[k]), k
