"""Process-group glue of the training CLIs: `python -m torch.distributed.run --nproc-per-node N -m nasrec_amd.main_train ...` runs one
process per GPU (backend "nccl" == RCCL over xGMI); without torchrun's environment the CLIs run single-process as before.

The reference has no distributed mode.  Under data parallelism every rank reads its own shards, the global batch
is `train_batch_size x world`, the path of a supernet step is drawn from an identically seeded np.random on every rank
(supernet.py:525-529 draws one path per step), and the step equals a single process at the global batch (nasrec_amd/parallel.py).

What keeps N replicas ONE model (each of these was a way to desynchronise silently):
  * `init_from_env` seeds torch (CPU + GPU generators) and np.random identically on every rank;
  * `broadcast_replica_state` sends rank 0's parameters, tables and Adagrad accumulators to everybody after the weights are
    initialised or loaded, and `assert_replicas_identical` checks a checksum of them before the first step;
  * `StepAgreement`: the ranks agree, one step ahead and off the compute stream, on whether EVERY rank holds a full batch for the
    next step — the epoch ends for all of them at the first step some rank cannot take (shards differ in length), so collectives
    never go unmatched;
  * `allreduce_grads` averages the gradients on the torch route (weight decay, Adam / SGD, frozen-parameter modes), where the fused
    engine step and its own exchange do not apply;
  * `any_rank` makes the NaN exit collective."""
import os

import torch
import torch.distributed as dist


def init_from_env(args=None):
    """-> (rank, world).  Must run before anything touches the GPU (a process that has initialised the GPU is never re-executed)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("NASREC_DIST_BACKEND", "nccl")  # "gloo": the CPU tests of this module
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args is not None and backend == "nccl":
        args.gpu = local
    seed = int(os.environ.get("NASREC_PATH_SEED", "0"))
    import numpy as np
    np.random.seed(seed)  # one path per step, the same on every rank
    torch.manual_seed(seed)  # constructor inits and init_weights draw the same numbers on every rank (CPU and GPU generators)
    return rank, world


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _replica_tensors(model):
    """every tensor that defines a replica: parameters, buffers, and the engine's Adagrad accumulators once they exist"""
    sharded = getattr(model, "_table_sharding", None) == "row"  # row-sharded tables: every rank owns DIFFERENT rows — not replica state
    out = [p.data for n, p in model.named_parameters() if not (sharded and n.startswith("_embedding."))] + [b.data for b in model.buffers()]
    eng = getattr(model, "_engine", None)
    if eng is not None:
        if getattr(eng, "flat_s", None) is not None:
            out.append(eng.flat_s)
        for t in (getattr(eng, "table_state", None) or []):
            out.append(t)
    return out


def broadcast_replica_state(model, src: int = 0):
    """rank `src`'s weights / tables / optimizer accumulators -> every rank (no-op in a single process)"""
    if world_info()[1] <= 1:
        return
    for t in _replica_tensors(model):
        dist.broadcast(t, src)


def replica_checksum(model) -> torch.Tensor:
    """[sum, sum of squares, count] over every replica tensor, in float64"""
    acc = None
    for t in _replica_tensors(model):
        v = t.detach().double()
        s = torch.stack([v.sum(), (v * v).sum(), torch.tensor(float(v.numel()), dtype=torch.float64, device=v.device)])
        acc = s if acc is None else acc + s.to(acc.device)
    return acc


def assert_replicas_identical(model):
    """raise unless every rank holds the same weights (checked before the first step of a data-parallel run)"""
    rank, world = world_info()
    if world <= 1:
        return
    mine = replica_checksum(model)
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError("data-parallel replicas differ before the first step (rank %d checksum %s, range %s .. %s): "
                           "broadcast_replica_state was not called after the weights were set" % (rank, mine.tolist(), lo.tolist(), hi.tolist()))


def allreduce_grads(model):
    """torch route under data parallelism: average every existing gradient over the ranks (dense tensors, bucketed).  The ranks run
    the same path, so the `grad is None` sets agree.  The tables' dense gradients travel too — correct and slow (2.16 GB on Criteo):
    the fused engine step (Adagrad, no weight decay) exchanges row gradients instead."""
    rank, world = world_info()
    if world <= 1:
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    bucket, size = [], 0
    LIMIT = 64 << 20  # elements

    def flush():
        if not bucket:
            return
        flat = torch._utils._flatten_dense_tensors(bucket) if len(bucket) > 1 else bucket[0].reshape(-1)
        dist.all_reduce(flat)
        flat.div_(world)
        if len(bucket) > 1:
            for g, f in zip(bucket, torch._utils._unflatten_dense_tensors(flat, bucket)):
                g.copy_(f)
        del bucket[:]

    for g in grads:
        if g.is_sparse:
            raise RuntimeError("sparse gradients are not exchanged on the torch route")
        if size + g.numel() > LIMIT and bucket:
            flush()
            size = 0
        bucket.append(g)
        size += g.numel()
    flush()


def any_rank(flag: bool, device=None) -> bool:
    """collective OR of a host flag (blocking; used at display intervals only)"""
    rank, world = world_info()
    if world <= 1:
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_side_group())
    return bool(int(t.item()))


_SIDE_GROUP = None


def _side_group():
    """one extra process group (its own communicator and stream) for the small control collectives — step agreement, the collective
    NaN flag — so that they never queue behind the gradient exchange of the default group.  Created collectively on first use:
    every rank reaches its first control collective at the same point of the program."""
    global _SIDE_GROUP
    if _SIDE_GROUP is None or _SIDE_GROUP[0] is not dist.group.WORLD:
        _SIDE_GROUP = (dist.group.WORLD, dist.new_group())
    return _SIDE_GROUP[1]


def _coll_device(device=None):
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device()) if device is None or not str(device).startswith("cuda") else torch.device(device)
    return torch.device("cpu")


class StepAgreement:
    """`post(ok)` for step k + 1 while step k runs, `take()` at the top of step k + 1: True iff EVERY rank said ok.  The tiny
    all-reduce(MIN) runs in a process group OF ITS OWN (ProcessGroupNCCL runs all collectives of one group on one internal stream in
    issue order: on the default group the agreement for step k + 1 would queue behind step k - 1's gradient all-reduce, and `take()`
    would wait for that step's backward) and is issued from a side stream, so reading its result does not wait for the training
    kernels queued on the compute stream.  Single process: the local flag."""

    def __init__(self, device=None):
        self.world = world_info()[1]
        self.pending = None
        self.dev = None
        self.stream = None
        self.group = None
        if self.world > 1:
            self.dev = _coll_device(device)
            self.group = _side_group()
            if self.dev.type == "cuda":
                self.stream = torch.cuda.Stream(self.dev)

    def post(self, ok: bool):
        if self.world <= 1:
            self.pending = bool(ok)
            return
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        else:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        self.pending = t

    def take(self) -> bool:
        p, self.pending = self.pending, None
        assert p is not None, "StepAgreement.take() without a post()"
        if isinstance(p, bool):
            return p
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                return bool(int(p.item()))
        return bool(int(p.item()))
