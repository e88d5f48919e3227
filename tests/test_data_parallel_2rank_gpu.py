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


def _inputs(z, meta, rows):
    """the golden batch, or (rows > 0) a seeded synthetic batch of that many samples over the golden network's tables: global batches
    above 256 samples take the per-(field, chunk) id half of the row dedup and the LDS-resident sums (csrc/dedup_bodies.h)"""
    import numpy as np
    if not rows:
        return z["int_x"], z["cat_x"], z["y"].reshape(-1)
    g = np.random.default_rng(7)
    Fd = z["int_x"].shape[1]
    int_x = g.standard_normal((rows, Fd)).astype(z["int_x"].dtype) * 0.5
    tabs = np.asarray(meta["tables"], dtype=np.int64)
    cat_x = (g.integers(0, 1 << 40, size=(rows, len(tabs))) % np.minimum(tabs, 97)[None, :]).astype(np.int64)  # (few distinct ids: duplicates within and across chunks and ranks)
    y = (g.random(rows) < 0.3).astype(z["y"].dtype)
    return int_x, cat_x, y


def _worker(rank, port, case, steps, pack, out, rows=0):
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
    int_x, cat_x, y = (torch.tensor(a).cuda() for a in _inputs(z, meta, rows))
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


@pytest.mark.parametrize("case,pack,rows", [("fixed_criteo_xlarge", 65536, 0), ("fixed_criteo_xlarge", 0, 0), ("fixed_kdd_autoctr", 65536, 0),
                                            ("supernet_xlarge_any", 0, 0), ("fixed_criteo_xlarge", 65536, 600), ("supernet_xlarge_any", 0, 300)])
def test_two_ranks_on_one_gpu_equal_one_process_at_the_global_batch(case, pack, rows):
    from helpers import GOLDEN, load_golden
    from test_parity_gpu import build_engine
    z, meta = load_golden(os.path.join(GOLDEN, case + ".npz"))
    steps = 3
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(port, case, steps, pack, out, rows), nprocs=WORLD, join=True)
    # one process, whole batch
    int_x, cat_x, y = (torch.tensor(a).cuda() for a in _inputs(z, meta, rows))
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


# ------------------------------------------------------------------------------------------------------------------
# paths="per-rank" on the REAL engine (round 6; the advisor's finding on round 5: `_per_rank_step` never launched the id half of the
# row dedup, so at global batches <= NASREC_DEDUP_SPLIT_MAX_B the optimizer read leaders / run lists of an earlier batch — the mode
# had only ever run with the CPU stand-in, whose `ids_half` is None).  Two ranks on one GPU, each with its own sampled path per step;
# expected: the fp64 oracle evaluating path r on half r, gradients summed over the ranks, clip + Adagrad over the union.
# ------------------------------------------------------------------------------------------------------------------
PER_RANK_CASE, PER_RANK_STEPS = "supernet_xlarge_any", 3


def _per_rank_choices(meta, same):
    import json
    import numpy as np
    from helpers import oracle_cfg
    from oracle import nasrec_oracle as O
    cfg = oracle_cfg(meta)
    out = []
    for r in range(WORLD):
        np.random.seed(977 + (0 if same else r))
        sampler = O.PathSampler(cfg, "default", "binomial-0.5")
        out.append([json.loads(json.dumps(sampler.sample(), default=lambda o: o.tolist() if hasattr(o, "tolist") else o.item()))
                    for _ in range(PER_RANK_STEPS)])
    return out


def _per_rank_worker(rank, port, same, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from helpers import GOLDEN, load_golden
    from nasrec_amd.parallel import DataParallelStep
    from test_parity_gpu import build_engine
    z, meta = load_golden(os.path.join(GOLDEN, PER_RANK_CASE + ".npz"))
    int_x, cat_x, y = (torch.tensor(a).cuda() for a in _inputs(z, meta, 0))
    Bl = int_x.shape[0] // WORLD
    sl = slice(rank * Bl, (rank + 1) * Bl)
    eng = build_engine(z, meta)
    dp = DataParallelStep(eng, None, Bl, clip=5.0, eps=1e-2, graph=False, paths="per-rank")
    choices = _per_rank_choices(meta, same)
    arena0 = None
    arena_sizes = []
    for s in range(PER_RANK_STEPS):
        dp.step(int_x[sl].contiguous(), cat_x[sl].contiguous(), y[sl].contiguous(), meta["lr"], choice=choices[rank][s])
        torch.cuda.synchronize()
    # a cached plan's arena must not grow with the number of steps it has run (the union's chunk table is allocated once per plan)
    plan = dp._last[1]
    arena = getattr(plan.cp, "arena", None)
    used = (lambda: int(arena.used_bytes()) if arena is not None else None)
    before = used()
    for _ in range(4):
        dp.step(int_x[sl].contiguous(), cat_x[sl].contiguous(), y[sl].contiguous(), 0.0, choice=choices[rank][PER_RANK_STEPS - 1])
    torch.cuda.synchronize()
    eng.check_indices()
    out[rank] = dict(params={k: v.cpu() for k, v in eng.state_dict().items()}, ids_half=dp.ids_half is not None, arena=(before, used()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("same", [True, False])
def test_per_rank_paths_on_the_engine_match_the_fp64_oracle(same):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import GOLDEN, load_golden, oracle_cfg, oracle_params
    from oracle import nasrec_oracle as O
    z, meta = load_golden(os.path.join(GOLDEN, PER_RANK_CASE + ".npz"))
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_per_rank_worker, args=(port, same, out), nprocs=WORLD, join=True)
    r0, r1 = out[0], out[1]
    assert r0["ids_half"], "a global batch this small takes the two-halves row dedup: the id half is the caller's launch"
    assert r0["arena"][0] == r0["arena"][1], "the plan's arena grew while a cached plan was re-run: %s" % (r0["arena"],)
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), "replicas differ: %s" % k
    # fp64 oracle: path r on half r, loss = mean over the GLOBAL batch
    cfg = oracle_cfg(meta)
    P = oracle_params(meta)
    choices = _per_rank_choices(meta, same)
    if not same:
        import json
        assert any(json.dumps(a) != json.dumps(b) for a, b in zip(choices[0], choices[1])), "the two ranks should train different paths"
    int_x, cat_x, y = (torch.tensor(a) for a in _inputs(z, meta, 0))
    int_x, y = int_x.double(), y.double().view(-1)
    Bg = int_x.shape[0]
    Bl = Bg // WORLD
    names = list(P.keys())
    state = {}
    for s in range(PER_RANK_STEPS):
        total = {}
        for r in range(WORLD):
            sl = slice(r * Bl, (r + 1) * Bl)
            leaves = {k: v.detach().requires_grad_(True) for k, v in P.items()}
            Pl = O.Params(P.dtype, frozen=True)
            Pl.update(leaves)
            logits = O.supernet_forward(Pl, cfg, int_x[sl], cat_x[sl], choices[r][s], num_embeddings=meta["tables"])
            loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), y[sl], reduction="sum") / Bg
            grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
            for k, g in zip(names, grads):
                if g is not None:
                    total[k] = total[k] + g if k in total else g.clone()
        norm = torch.sqrt(sum(g.pow(2).sum() for g in total.values()))
        coef = min(1.0, 5.0 / (float(norm) + 1e-6))
        with torch.no_grad():
            for k, g in total.items():
                st = state.setdefault(k, torch.zeros_like(P[k]))
                O.adagrad_step_(P[k], g * coef, st, meta["lr"], 1e-2)
    bad = []
    for k, v in P.items():
        got = r0["params"][k].double().reshape(v.shape)
        scale = max(1.0, float(v.abs().max()))
        err = float((got - v).abs().max())
        if err > 1e-4 * scale:
            bad.append((k, err))
    assert not bad, bad[:8]
