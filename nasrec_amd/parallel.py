"""Data-parallel training step: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The path shards over the batch (samples are independent; LayerNorm is per row — SURVEY §8e).  Per step, after the
local forward/backward:
  * dense gradients: ONE all-reduce (sum) of the flat gradient arena; the 1/(B·world) factor is already folded into
    dlogits, so the sum equals the gradient of the mean loss over the global batch;
  * embedding gradients: all-gather of (ids [B,Fs] int64, per-sample row gradients [B,Fs,16]) — B_global·Fs·72 bytes,
    instead of all-reducing 2.16 GB of dense table gradient — then every rank runs the same row-sparse
    dedup + clip + Adagrad over the global batch, so the replicated tables stay bit-identical across ranks;
  * the global-norm clip coefficient comes out identical on every rank because it is computed from identical data.
Overlap (DataParallelStep.step): the three collectives are issued asynchronously on RCCL's stream as soon as their
inputs exist — the ids right after staging (hidden under the whole forward/backward), the row gradients when the
backward chain reaches the embedding stem (hidden under the weight-gradient products, which the plan parks at the end
of the backward program), the dense arena after the last product — and the compute stream only waits for them in front
of the optimizer launches.
The reference has no distributed code at all; equivalence target = a single process at the global batch.
"""
from typing import Optional

import torch
import torch.distributed as dist

from .engine import Program, SupernetEngine


def all_gather_rows(out: torch.Tensor, local: torch.Tensor):
    """out[r*n:(r+1)*n] = local of rank r (n = local.numel()); works on nccl and gloo."""
    try:
        dist.all_gather_into_tensor(out.view(-1), local.reshape(-1))
    except (RuntimeError, NotImplementedError):
        n = local.numel()
        chunks = [out.view(-1)[r * n:(r + 1) * n] for r in range(dist.get_world_size())]
        dist.all_gather(chunks, local.reshape(-1))


def all_gather_rows_async(out: torch.Tensor, local: torch.Tensor):
    """all_gather_rows without blocking the compute stream; returns the work handle (None if it completed inline)"""
    try:
        return dist.all_gather_into_tensor(out.view(-1), local.reshape(-1), async_op=True)
    except (RuntimeError, NotImplementedError):
        all_gather_rows(out, local)
        return None


def exchange_gradients(flat_g: torch.Tensor, cat_local: torch.Tensor, sg_local: torch.Tensor, cat_all: torch.Tensor, sg_all: torch.Tensor):
    """the only cross-rank traffic of a step (see module docstring), issued back to back"""
    dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
    all_gather_rows(cat_all, cat_local)
    all_gather_rows(sg_all, sg_local)


class _Holder:
    pass


class DataParallelStep:
    def __init__(self, engine: SupernetEngine, choice, B_local: int, clip: Optional[float] = 5.0, eps: float = 1e-2, graph: bool = True,
                 force_exchange: bool = False):
        """force_exchange: take the multi-rank code path (all-reduce + all-gather + global-batch optimizer) even in a
        single-rank process group — lets one GPU exercise exactly what N GPUs run."""
        self.engine = engine
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.B = B_local
        self.graph = graph and engine.cfg.fixed
        self.exchange = self.world > 1 or (force_exchange and dist.is_initialized())
        if not self.exchange:
            self.cp = engine.compile(choice, B_local, True, clip, eps, graph=self.graph)
            self.choice = choice
            self.clip, self.eps = clip, eps
            return
        self.cp = engine.compile(choice, B_local, True, clip, eps, graph=False, grad_scale=1.0 / (B_local * self.world))
        with torch.cuda.stream(engine.stream):
            Bg = B_local * self.world
            self.cat_all = torch.zeros(Bg, engine.Fs, dtype=torch.int64, device=engine.device)
            self.sg_all = torch.zeros(Bg * engine.Fs * 16, dtype=torch.float32, device=engine.device)
            self.holder = _Holder()
            self.opt = Program(engine._optimizer_descs(self.holder, Bg, self.cat_all, self.sg_all, clip, eps))
            # forward + backward chain | parked weight-gradient products: the row-gradient all-gather runs under the latter
            cut = self.cp.bwd_tail_start
            self.fb = Program(self.cp.fwd.descs + self.cp.bwd.descs[:cut])
            self.tail = Program(self.cp.bwd.descs[cut:]) if cut < len(self.cp.bwd.descs) else None
            if self.graph:
                self.fb.capture(engine.stream.cuda_stream)
                if self.tail is not None:
                    self.tail.capture(engine.stream.cuda_stream)
                self.opt.capture(engine.stream.cuda_stream)
        engine.stream.synchronize()

    def step(self, int_x, cat_x, y, lr: float):
        eng = self.engine
        if not self.exchange:
            return eng.train_step(int_x, cat_x, y, lr, self.choice, self.clip, self.eps, graph=self.graph)
        sp = eng._sp()
        run = (lambda prog: prog.replay(sp)) if self.graph else (lambda prog: prog.run(sp))
        eng._stage_inputs(sp, self.cp, int_x, cat_x, y, lr)
        pending = [all_gather_rows_async(self.cat_all, self.cp.cat_x)]
        run(self.fb)
        pending.append(all_gather_rows_async(self.sg_all, self.cp.sparse0.grad_tensor()))
        if self.tail is not None:
            run(self.tail)
        pending.append(dist.all_reduce(eng.flat_g, op=dist.ReduceOp.SUM, async_op=True))
        for w in pending:
            if w is not None:
                w.wait()  # nccl: the compute stream waits for the collective (no host block)
        run(self.opt)
        return self.cp.loss
