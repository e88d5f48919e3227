"""Data-parallel training step: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The path shards over the batch (samples are independent; LayerNorm is per row — SURVEY §8e).  Per step, after the
local forward/backward:
  * dense gradients: all-reduce (sum) of the flat gradient arena; the 1/(B·world) factor is already folded into dlogits, so the
    sum equals the gradient of the mean loss over the global batch.  Fixed sub-network (8.9 MB, of which the backward's live set
    writes about half): the joint forward + backward launch list is cut into NASREC_DP_SEGMENTS pieces where finished gradient
    has piled up (`gradient_ready_index`: last writer of every parameter from the launches' own footprints), and each piece is
    followed by the all-reduce of the ranges final after it.  Supernet (up to 686 MB of parameters, of which a sampled path touches a fraction): the backward program is cut at
    the block boundaries, and as soon as block i's gradients are complete the arena ranges of the parameters THIS PATH trained
    in block i go out as one asynchronous all-reduce each — buckets in reverse block order, overlapped with the rest of the
    backward; parameters the path did not touch are zero on every rank and are not sent at all;
  * embedding gradients: all-gather of (ids [B,Fs] int64, per-sample row gradients [B,Fs,16]) — B_global·Fs·72 bytes,
    instead of all-reducing 2.16 GB of dense table gradient — then every rank runs the same row-sparse
    dedup + clip + Adagrad over the global batch, so the replicated tables stay bit-identical across ranks;
  * the global-norm clip coefficient comes out identical on every rank because it is computed from identical data.
Fixed sub-networks (batch 256, a 0.3 ms step): work handles cost ~40 us of host time per collective, more than the step can hide,
so the WHOLE exchange step — launches, collectives on RCCL's stream and the event edges between the two streams — is captured
once into a graph (`DataParallelStep._capture`); a step is the staging launch plus one graph launch.
Path sampling under DP: the reference draws ONE path per step from the global `np.random` stream (supernet.py:525-529); all
ranks share the seed, hence the same path, hence a step that equals a single process at the global batch (`choice=` of
`DataParallelStep.step`).  The collectives are asynchronous: the ids go out right after staging, the row gradients when the
backward chain reaches the embedding stem, and the compute stream waits for them only in front of the optimizer launches.
The reference has no distributed code at all; equivalence target = a single process at the global batch.

`DataParallelStep` drives an engine through a small protocol (`dp_plan`, `dp_optimizer`, `flat_g`, `Fs`, `device`, `cfg.fixed`),
which `SupernetEngine` implements below through `EngineDP`; tests/test_data_parallel_cpu.py runs the SAME step code under gloo
with a CPU stand-in that implements the protocol with the oracle.
"""
import json
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

BUCKET_GAP = 16384  # floats: arena ranges closer than this are sent as one all-reduce (fewer, larger collectives)


class Collectives:
    """The step's cross-rank calls behind one object (tools/dp_overhead.py wraps it to leave one kind out and time the rest; the
    product path never does).  `real_at_world1`: a single-rank group still goes through the library — RCCL's all-gather and
    all-reduce kernels run (and are captured into the step graph) exactly as on N ranks, which is what a one-GPU box can verify of the
    N-GPU step; off, a one-rank gather is a copy kernel (the cheapest correct thing)."""

    def __init__(self, real_at_world1: bool = False):
        self.real = bool(real_at_world1)

    def all_gather(self, out: torch.Tensor, local: torch.Tensor, async_op: bool = False):
        """out[r*n:(r+1)*n] = local of rank r (n = local.numel()); nccl and gloo.  Returns the work handle of an asynchronous call
        (None when it completed inline)."""
        if dist.get_world_size() == 1 and not self.real:
            # a gather over one rank is a copy: as a kernel on the calling stream, not as the library's device-to-device memcpy (which a
            # captured graph replays as a memcpy node: ~30 us each on this stack, the whole 62 us the single-rank exchange step cost)
            out.view(-1).copy_(local.reshape(-1))
            return None
        try:
            w = dist.all_gather_into_tensor(out.view(-1), local.reshape(-1), async_op=async_op)
        except (RuntimeError, NotImplementedError):
            n = local.numel()
            chunks = [out.view(-1)[r * n:(r + 1) * n] for r in range(dist.get_world_size())]
            w = dist.all_gather(chunks, local.reshape(-1), async_op=async_op)
        return w if async_op else None

    def all_reduce(self, t: torch.Tensor, async_op: bool = False):
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)
        return w if async_op else None


_DEFAULT = Collectives()


def all_gather_rows(out: torch.Tensor, local: torch.Tensor):
    """out[r*n:(r+1)*n] = local of rank r (n = local.numel()); works on nccl and gloo."""
    _DEFAULT.all_gather(out, local)


def all_gather_rows_async(out: torch.Tensor, local: torch.Tensor):
    """all_gather_rows without blocking the compute stream; returns the work handle (None if it completed inline)"""
    return _DEFAULT.all_gather(out, local, async_op=True)


def exchange_gradients(flat_g: torch.Tensor, cat_local: torch.Tensor, sg_local: torch.Tensor, cat_all: torch.Tensor, sg_all: torch.Tensor):
    """the only cross-rank traffic of a step (see module docstring), issued back to back"""
    dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
    all_gather_rows(cat_all, cat_local)
    all_gather_rows(sg_all, sg_local)


def coalesce_ranges(ranges: List[Tuple[int, int]], gap: int = BUCKET_GAP) -> List[Tuple[int, int]]:
    """[(offset, numel)] -> merged, ascending; neighbours closer than `gap` floats become one range"""
    out = []
    for off, n in sorted(ranges):
        if out and off <= out[-1][0] + out[-1][1] + gap:
            end = max(out[-1][0] + out[-1][1], off + n)
            out[-1] = (out[-1][0], end - out[-1][0])
        else:
            out.append((off, n))
    return out


DP_SEGMENTS = int(__import__("os").environ.get("NASREC_DP_SEGMENTS", "4"))
# Dense gradients that become final in the LAST pieces of the backward cannot hide behind it: up to this many floats of them ride in the
# row-gradient all-gather instead of an all-reduce of their own (ONE exposed collective per step), and every rank adds the W copies in
# rank order.  0 = every piece is all-reduced.
_DEBUG = __import__("os").environ.get("NASREC_DP_DEBUG", "0") == "1"
# where the id-only half of the global batch's row dedup runs in a data-parallel step (A/B knob): side = on a side stream behind the ids
# all-gather; chain = the gather itself is issued from that side stream; main = on the compute stream in front of the optimizer (exposed)
# (measured on one rank, profiles/r05_dp_overhead.txt: a side branch in the captured step costs ~40 us — the graph runs parallel branches
# through separate hardware queues — against 8 us for the kernel itself on the compute stream: main is the default)
_IDS_MODE = __import__("os").environ.get("NASREC_DP_IDS_MODE", "main")
# NASREC_DP_LATE_IDS=1 (with NASREC_DP_SEGMENTS=1: the LINEAR form of the exchange — nothing beside the forward / backward, every
# collective behind the backward's last launch): for stacks on which a branch in the captured step costs more than the transfer it hides
_LATE_IDS = __import__("os").environ.get("NASREC_DP_LATE_IDS", "0") == "1"
PACK_TAIL_FLOATS = int(__import__("os").environ.get("NASREC_DP_PACK_TAIL", "65536"))


def gradient_ready_index(eng, descs):
    """{dense parameter name: index of the LAST launch of `descs` that writes its gradient}; parameters no launch writes are
    left out.  Read off the launches' own write footprints (schedule.desc_io — the model the level scheduler trusts); a launch of a
    kind the model does not know counts as writing every gradient."""
    import bisect
    from . import _lib as L
    from . import schedule as S
    names = sorted(eng.offsets, key=lambda n: eng.offsets[n])
    starts = [eng.offsets[n] for n in names]
    base, total = eng.flat_g.data_ptr(), eng.flat_numel
    ready = {}
    for i, d in enumerate(descs):
        if isinstance(d, L.WorklistDesc):
            nodes = d.nodes
        elif d.kind == L.OP_SPLITK_EPILOGUES:
            nodes = [S.Node(d.g[q], "epi") for q in range(d.n)]
        else:
            nodes = [S.Node(d)]
        for nd in nodes:
            if nd.writes is None:
                for n in names:
                    ready[n] = i
                continue
            for w in nd.writes:
                lo, hi = (w.ptr - base) // 4, (w.end - base + 3) // 4
                if hi <= 0 or lo >= total:
                    continue
                j = max(bisect.bisect_right(starts, lo) - 1, 0)
                while j < len(names) and starts[j] < hi:
                    if starts[j] + eng.params[names[j]].numel() > lo:
                        ready[names[j]] = i
                    j += 1
    return ready


def merge_over_unwritten(eng, order, mine, ready):
    """arena ranges covering the parameters `mine` (one piece's finished gradients) with as few ranges as possible: two of them
    become one range when everything between them is alignment padding or gradients NO launch of the step writes (zero on every
    rank, now and for ever: summing zeros changes nothing) — never across a parameter another piece finishes.  Every collective
    costs a launch and a ring latency; a fixed sub-network's piece is one or two ranges this way."""
    out, cur = [], None
    for n in order:
        if n in mine:
            lo, hi = eng.offsets[n], eng.offsets[n] + eng.params[n].numel()
            cur = [cur[0], hi] if cur is not None else [lo, hi]
            if out and out[-1][2]:
                out[-1] = cur + [True]
            else:
                out.append(cur + [True])
        elif n in ready:  # somebody else's gradient: the open range ends in front of it
            cur = None
            if out:
                out[-1][2] = False
    return [(lo, hi - lo) for lo, hi, _ in out]


def cut_segments(ready, n_launches, numel, nseg):
    """[(end index, [names final after descs[:end]])]: at most `nseg` pieces covering all launches; a cut is placed behind a launch
    once a 1/nseg share of the gradient bytes has become final since the previous cut"""
    by_idx = {}
    for n, i in ready.items():
        by_idx.setdefault(i, []).append(n)
    left = sum(numel[n] for n in ready)
    pieces, names, acc = [], [], 0
    idxs = sorted(by_idx)
    for pos, i in enumerate(idxs):
        names += by_idx[i]
        acc += sum(numel[n] for n in by_idx[i])
        share = max(left // max(nseg - len(pieces), 1), 1)  # an equal share of what is still to come (one huge parameter does not use up the cuts)
        if acc >= share and pos + 1 < len(idxs) and len(pieces) < nseg - 1:
            pieces.append((i + 1, names))
            left -= acc
            names, acc = [], 0
    pieces.append((n_launches, names))
    return pieces


class DPPlan:
    """What one (choice, local batch) looks like to the data-parallel step."""
    stage: Callable      # (int_x, cat_x, y, lr) -> None: batch into the plan's static buffers
    forward: Callable    # () -> None
    segments: List       # [(run, [(offset, numel), ...])]: backward pieces, each followed by the arena ranges complete after it
    cat_local: torch.Tensor
    sparse_grad: torch.Tensor
    loss: torch.Tensor


def tail_pieces(segments, budget: int) -> int:
    """index k such that the ranges of segments[k:] (the LAST pieces of the backward: nothing is left to hide them behind) hold at
    most `budget` floats in all; len(segments) when nothing fits.  The first piece is never packed when there are several (it is
    the one that overlaps)."""
    k, acc = len(segments), 0
    while k > (1 if len(segments) > 1 else 0):
        n = sum(m for _, m in segments[k - 1][1])
        if acc + n > budget:
            break
        acc += n
        k -= 1
    return k


class DataParallelStep:
    def __init__(self, engine, choice, B_local: int, clip: Optional[float] = 5.0, eps: float = 1e-2, graph: Optional[bool] = True,
                 force_exchange: bool = False, real_collectives: bool = False, paths: str = "shared"):
        """choice: the fixed sub-network's choice, or None for a weight-sharing supernet (the path then comes with every step).
        force_exchange: take the multi-rank code path (all-reduce + all-gather + global-batch optimizer) even in a single-rank
        process group — lets one GPU exercise exactly what N GPUs run.
        real_collectives: with it, a single-rank group goes through the communication library too (RCCL's all-gather / all-reduce
        kernels inside the captured step) instead of the copy a one-rank gather amounts to — adds work, for tests and measurement.
        paths: "shared" (default) = every rank trains the SAME sampled path per step (shared np.random seed): the reference's
        one-path-per-step semantics (supernet.py:525-529) at the global batch.  "per-rank" (weight-sharing supernets only; an
        EXTENSION of the reference) = every rank passes its OWN path to step(): the choices are exchanged first (one small
        all-gather), the dense gradients are all-reduced over the UNION of the ranks' parameter ranges — a parameter's gradient is the
        sum over the ranks whose path used it, the others contribute zeros — and clip + Adagrad run over that union on every rank
        (replicas stay bit-identical); parameters no rank's path touched keep value and state, as `grad is None` does in the reference."""
        self.engine = engine
        self.dp = engine if hasattr(engine, "dp_plan") else EngineDP(engine)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.B = B_local
        self.fixed = bool(engine.cfg.fixed)
        self.exchange = self.world > 1 or (force_exchange and dist.is_initialized())
        # graph = None: the exchange step is always captured (work handles cost more than a 0.3 ms step hides); the plain step of one
        # rank replays or launches, whichever is faster for its plan (engine.prefers_graph)
        if graph is None:
            graph = self.fixed and (self.exchange or not hasattr(engine, "prefers_graph") or engine.prefers_graph(B_local))
        self.graph = bool(graph and self.fixed)
        self.choice, self.clip, self.eps = choice, clip, eps
        self.coll = Collectives(real_collectives)
        assert paths in ("shared", "per-rank"), paths
        self.per_rank = paths == "per-rank"
        assert not (self.per_rank and self.fixed), "a fixed sub-network has one path"
        self._last = None
        self._plans = {}
        self._last_key = None
        if not self.exchange:
            if self.fixed:
                self.cp = engine.compile(choice, B_local, True, clip, eps, graph=self.graph)
            return
        Bg = B_local * self.world
        dev = engine.device
        gdt = getattr(engine, "grad_dtype", torch.float32)
        self.rows_n = B_local * engine.Fs * 16
        self.cat_all = torch.zeros(Bg, engine.Fs, dtype=torch.int64, device=dev)
        # this rank's share of the row-gradient all-gather: [B, Fs, 16] row gradients | the dense gradients of the backward's last
        # pieces (fixed sub-networks: `tail`); the receive buffer holds the W shares one after the other
        cap = min(PACK_TAIL_FLOATS, int(engine.flat_numel)) if self.fixed else 0  # (never more than the arena holds)
        self.sg_send = torch.zeros(self.rows_n + cap, dtype=gdt, device=dev)
        self.sg_recv = torch.zeros(self.world * (self.rows_n + cap), dtype=gdt, device=dev)
        self.tail_n = None  # floats of packed tail per rank: fixed by the first plan (the optimizer's row layout depends on it)
        self.opt = None
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        if self.fixed:
            self.cp = self._plan(choice)

    # -------------------------------------------------------------------------------------------------------------
    def _plan(self, choice) -> DPPlan:
        if self.fixed and self._last_key is not None and self._last_key[0] is choice:  # same object as last step: skip the JSON key
            key = self._last_key[1]
        else:
            # keyed on CONTENT: a caller that rebuilds an equal choice dict every step must not re-compile (and re-capture) every step
            key = json.dumps(choice, sort_keys=True, default=_jsonable)
            self._last_key = (choice, key)
        hit = self._plans.get(key)
        if hit is not None and getattr(getattr(hit, "cp", None), "evicted", False):
            hit = None  # the engine recycled this plan's slot for another path
        if hit is None:
            if len(self._plans) >= 8:
                self._plans.pop(next(iter(self._plans)))
            kw = {}
            if self.fixed and self.sg_send.numel() > self.rows_n and getattr(self.dp, "accepts_row_grad_out", False):
                kw["row_grad_out"] = self.sg_send[:self.rows_n]  # the backward writes the row gradients where the gather sends them from
            hit = self.dp.dp_plan(choice, self.B, 1.0 / (self.B * self.world), self.clip, self.eps, False, **kw)
            self._layout(hit)
            self._plans[key] = hit
        return hit

    def _layout(self, plan):
        """which dense-gradient ranges ride in the row-gradient all-gather (plan.tail, plan.first_packed), and the optimizer over
        the resulting receive-buffer layout (built with the first plan)"""
        segs = plan.segments
        plan.tail, plan.first_packed = [], len(segs)
        if self.fixed and PACK_TAIL_FLOATS > 0:
            k = tail_pieces(segs, self.sg_send.numel() - self.rows_n)
            plan.first_packed = k
            plan.tail = [rg for _, ranges in segs[k:] for rg in ranges]
        tail_n = sum(n for _, n in plan.tail)
        if self.tail_n is None:
            self.tail_n = tail_n
            stride = self.rows_n + tail_n
            kw = {"rank_layout": (self.B, stride)} if tail_n else {}
            cpl = getattr(plan, "cp", None)
            if self.fixed and getattr(self.dp, "accepts_chunk_table", False) and getattr(cpl, "chunk_tab", None) is not None:
                kw["chunk_table"] = (cpl.chunk_tab, cpl.nchunks)  # norm + Adagrad over the ranges the sub-network's backward reaches
            self.recv = self.sg_recv[:self.world * stride]
            self.opt = self.dp.dp_optimizer(self.B * self.world, self.cat_all, self.recv, self.clip, self.eps, False, **kw)  # (graph: the whole exchange step is captured as one, _capture)
            self.ids_half = getattr(self.dp, "dp_dedup_ids", lambda: None)()
            self.tail_ops = None
            if tail_n:
                mk = getattr(self.dp, "dp_tail_ops", None)
                self.tail_ops = mk(self.sg_send, self.recv, self.rows_n, plan.tail, self.world) if mk is not None else \
                    _torch_tail_ops(self.engine.flat_g, self.sg_send, self.recv, self.rows_n, plan.tail, self.world)
        assert tail_n == self.tail_n, "every plan of one DataParallelStep packs the same number of dense-gradient floats"

    def step(self, int_x, cat_x, y, lr: float, choice=None):
        eng = self.engine
        choice = choice if choice is not None else self.choice
        if not self.exchange:
            loss = eng.train_step(int_x, cat_x, y, lr, choice, self.clip, self.eps, graph=self.graph)
            self._last = ("plain", choice, int(int_x.shape[0]))
            return loss
        union = None
        if self.per_rank:
            # the other ranks' paths first (their parameter ranges come from plans of their own, compiled into the engine's few plan
            # slots): this rank's plan is compiled LAST, so it is the one that stays resident
            spans = []
            for r, ch in enumerate(self._all_choices(choice)):
                if r != (dist.get_rank() if dist.is_initialized() else 0):
                    spans += list(self.dp.dp_path_spans(ch, self.B, 1.0 / (self.B * self.world), self.clip, self.eps))
            union = spans
        plan = self._plan(choice)
        plan.stage(int_x, cat_x, y, lr)
        self._last = ("dp", plan)
        if self.per_rank:
            self._per_rank_step(plan, coalesce_ranges(union + list(self.dp.dp_path_spans(choice, self.B, 1.0 / (self.B * self.world), self.clip, self.eps)), gap=0))
            return plan.loss
        if self.graph:
            # Batch-256 regime: the step is a few hundred microseconds and the HOST is what the exchange competes with (a work handle
            # costs ~40 us of host time per collective: 757 us against 609 us for the plain step in round 2).  So the whole exchange —
            # collectives on RCCL's stream, the event edges between it and the compute stream, every launch of the step — is
            # captured ONCE into a graph; a step is the staging launch + one graph launch, and the overlap costs the host nothing.
            g = getattr(plan, "step_graph", None)
            if g is None:
                g = plan.step_graph = self._capture(plan)
            if g is not False:
                g.replay()
                return plan.loss
        self._exchange_step(plan)
        return plan.loss

    def _exchange_step(self, plan):
        """forward, backward in segments, the exchange under it, the optimizer over the global batch — the same sequence for fixed
        sub-networks and sampled paths, eager (work handles) or under graph capture.
          ids          all-gather right after staging: hidden under the whole forward / backward.  The id-only half of the global
                       batch's row dedup (leaders, duplicate lists: csrc/dedup_bodies.h) runs where NASREC_DP_IDS_MODE says: "main"
                       (the default) = on the compute stream right in front of the optimizer, exposed (a fork to a second stream inside a
                       captured graph cost ~40 us on this stack, more than the launch); "side" / "chain" = on a side stream beside the
                       forward, off the step's tail;
          dense grads  one all-reduce per (segment, arena range) as soon as the segment that completes the range has been
                       enqueued — RCCL's stream waits for that point of the compute stream only, the rest of the backward runs
                       beside the transfer.  The LAST pieces' ranges (nothing left to hide them behind; `plan.tail`, a few tens of
                       KB) are not all-reduced: they ride behind the row gradients in the one all-gather and every rank adds the
                       W copies in rank order (same bits on every rank);
          row grads    all-gather when the backward has reached the embedding stem (its last launch): the step's ONE exposed collective;
        the compute stream waits for all of them in front of the optimizer launches, nowhere else."""
        flat_g = self.engine.flat_g
        coll = self.coll
        pending = []
        ids_done = None
        mode = _IDS_MODE if (self.ids_half is not None and self._side is not None) else ("main" if self.ids_half is not None else None)
        if mode == "chain":
            # the ids all-gather is ISSUED from the side stream (a blocking call there: the side stream waits for the library's stream),
            # the id half right behind it — one chain beside the compute stream instead of two forks
            main = torch.cuda.current_stream(self.engine.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                coll.all_gather(self.cat_all, plan.cat_local)
                self.ids_half()
                ids_done = torch.cuda.Event()
                ids_done.record(self._side)
            w_ids = None
        elif _LATE_IDS:
            w_ids = None  # (issued with the row gradients, behind the backward: no branch beside the forward)
        else:
            w_ids = coll.all_gather(self.cat_all, plan.cat_local, async_op=True)
        if mode == "side":
            main = torch.cuda.current_stream(self.engine.device)
            self._side.wait_stream(main)  # (the previous step's optimizer has read the lists this launch rewrites)
            with torch.cuda.stream(self._side):
                if w_ids is not None:
                    w_ids.wait()
                self.ids_half()
                ids_done = torch.cuda.Event()
                ids_done.record(self._side)
        elif w_ids is not None:
            pending.append(w_ids)
        plan.forward()
        for k, (run, ranges) in enumerate(plan.segments):
            run()
            if k >= plan.first_packed:
                continue
            for off, n in ranges:
                pending.append(coll.all_reduce(flat_g[off:off + n], async_op=True))
        if self.tail_n:
            send = self.sg_send[:self.rows_n + self.tail_n]
            if plan.sparse_grad.data_ptr() != send.data_ptr():
                send[:self.rows_n].copy_(plan.sparse_grad.view(-1))
            self.tail_ops[0]()  # dense gradients of the last pieces -> behind the rows
        else:
            send = plan.sparse_grad
        if _LATE_IDS and mode != "chain":
            pending.append(coll.all_gather(self.cat_all, plan.cat_local, async_op=True))
        pending.append(coll.all_gather(self.recv, send, async_op=True))
        for w in pending:
            if w is not None:
                w.wait()  # nccl: the compute stream waits for the collective (no host block)
        if ids_done is not None:
            torch.cuda.current_stream(self.engine.device).wait_event(ids_done)
        if mode == "main":
            self.ids_half()  # (no side stream — CPU stand-ins — or NASREC_DP_IDS_MODE=main: in front of the optimizer, exposed)
        if self.tail_n:
            self.tail_ops[1]()  # flat_g[tail ranges] = sum over ranks, in rank order
        self.opt(plan)
        if _DEBUG and getattr(plan, "never_written", None) and not torch.cuda.is_current_stream_capturing():
            for n in plan.never_written:
                assert float(self.engine.grads[n].abs().max()) == 0.0, "gradient of %s is written by nothing and must stay zero" % n

    # -------------------------------------------------------------------------------------------------------------
    CHOICE_BYTES = 16384

    def _all_choices(self, choice):
        """every rank's path of this step, in rank order (one all-gather of the JSON text, padded)"""
        raw = json.dumps(choice, sort_keys=True, default=_jsonable).encode()
        assert len(raw) < self.CHOICE_BYTES - 8, "choice too long for the exchange buffer"
        dev = self.engine.device
        buf = torch.zeros(self.CHOICE_BYTES, dtype=torch.uint8)
        buf[:8] = torch.tensor(list(len(raw).to_bytes(8, "little")), dtype=torch.uint8)
        buf[8:8 + len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
        buf = buf.to(dev)
        out = torch.zeros(self.world * self.CHOICE_BYTES, dtype=torch.uint8, device=dev)
        self.coll.all_gather(out, buf)
        host = out.cpu().view(self.world, self.CHOICE_BYTES)  # (the host needs them: this mode synchronises once per step)
        res = []
        for r in range(self.world):
            n = int.from_bytes(bytes(host[r, :8].tolist()), "little")
            res.append(json.loads(bytes(host[r, 8:8 + n].tolist()).decode()))
        return res

    def _per_rank_step(self, plan, union):
        """paths="per-rank": own path forward / backward; all-reduce and optimizer over the union of the ranks' parameter ranges"""
        flat_g = self.engine.flat_g
        coll = self.coll
        pending = [coll.all_gather(self.cat_all, plan.cat_local, async_op=True)]
        # gradients of the union's parameters that THIS rank's path does not write must be zero on this rank (the plan's own
        # zero_grad covers its path only): zero the union first, the plan then writes its share
        for off, n in union:
            flat_g[off:off + n].zero_()
        plan.forward()
        for run, _ in plan.segments:
            run()
        for off, n in union:
            pending.append(coll.all_reduce(flat_g[off:off + n], async_op=True))
        pending.append(coll.all_gather(self.recv, plan.sparse_grad, async_op=True))
        for w in pending:
            if w is not None:
                w.wait()
        if self.ids_half is not None:
            # global batch <= NASREC_DEDUP_SPLIT_MAX_B: the optimizer program is [OPT_REDUCE2, OPT_APPLY] only and reads the leaders /
            # run lists the id half writes for THIS batch — it is the caller's launch (as in _exchange_step), behind the ids all-gather
            self.ids_half()
        self.opt(plan, spans=union)
        plan.union = union

    def _capture(self, plan):
        """the exchange step of a fixed sub-network as ONE graph (torch.cuda.CUDAGraph: ProcessGroupNCCL's collectives are capturable,
        and the engine launches on whatever stream is current — inside the capture that is the capturing stream).  Returns False when
        the platform refuses (the step then runs eagerly, same results)."""
        dev = self.engine.device
        if dev.type != "cuda":
            return False
        # communicators are created lazily by the first collective of each kind, which must not happen inside a capture
        w0 = torch.zeros(8, device=dev)
        w1 = torch.zeros(8 * self.world, device=dev)
        dist.all_reduce(w0)
        Collectives(True).all_gather(w1, w0)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._exchange_step(plan)
        except Exception as e:  # noqa: BLE001
            import warnings
            warnings.warn("data-parallel step: graph capture of the exchange failed (%s); running it eagerly" % (e,))
            torch.cuda.synchronize(dev)
            return False
        torch.cuda.synchronize(dev)
        return g

    def last_loss(self):
        if self._last[0] == "dp":
            return self._last[1].loss
        return self.last_plan().loss

    def last_plan(self):
        """the engine's compiled plan of the most recent step (bench.py reads its descriptors)"""
        if self._last[0] == "dp":
            return self._last[1].cp
        _, choice, B = self._last
        return self.engine.compile(choice, B, True, self.clip, self.eps, graph=self.graph)


def _torch_tail_ops(flat_g, send, recv, rows_n, tail, world):
    """pack / unpack of the packed tail with plain tensor operations (any device; the engine has one launch for each: EngineDP.dp_tail_ops)"""
    stride = rows_n + sum(n for _, n in tail)

    def pack():
        pos = rows_n
        for off, n in tail:
            send[pos:pos + n].copy_(flat_g[off:off + n])
            pos += n

    def unpack():
        R = recv.view(world, stride)
        pos = rows_n
        for off, n in tail:
            acc = R[0, pos:pos + n].clone()
            for r in range(1, world):  # rank order
                acc += R[r, pos:pos + n]
            flat_g[off:off + n].copy_(acc)
            pos += n

    return pack, unpack


class EngineDP:
    """SupernetEngine behind the data-parallel protocol (programs, hipGraph segments, arena ranges per block)."""
    accepts_row_grad_out = True
    accepts_chunk_table = True

    def __init__(self, engine):
        self.engine = engine
        self._holder = None

    def dp_optimizer(self, Bg, cat_all, sg_all, clip, eps, graph, rank_layout=None, chunk_table=None):
        from .engine import Program
        eng = self.engine
        with torch.cuda.stream(eng.stream):
            holder = self._holder = _Holder()
            if chunk_table is not None:
                holder.chunk_tab, holder.nchunks = chunk_table
            eng._ensure_table_state()
            prog = Program(eng._optimizer_descs(holder, Bg, cat_all, sg_all, clip, eps, rank_layout=rank_layout))
            prog.holder = holder
            if graph:
                prog.capture(eng.stream.cuda_stream)
        eng.stream.synchronize()
        if eng.cfg.fixed:
            return (lambda plan: prog.replay(eng._sp())) if graph else (lambda plan: prog.run(eng._sp()))
        shared = ("leader", "gsum", "emb_partial", "dense_partial", "dd_order", "dd_lists", "dd_counts", "dd_heads", "emb_partial2")

        def run(plan, spans=None):
            if spans is not None:
                # paths="per-rank": norm and Adagrad over the UNION of the ranks' parameter ranges — a chunk table of its own, written
                # by the program's first launches (stream-ordered, like the plan's)
                from . import plan as P
                flat = P.path_chunks(spans)
                h = _Holder()
                for k in shared:
                    if hasattr(holder, k):
                        setattr(h, k, getattr(holder, k))
                # ONE table per plan, reused by every step (the union changes with the other ranks' paths: the table is rewritten in
                # stream order by the program's first launches) and regrown only when a larger union arrives — never out of the plan's
                # bump arena, which lives as long as the cached plan and would grow by a table per step
                tab = getattr(plan, "union_tab", None)
                if tab is None or tab.numel() < len(flat):
                    tab = plan.union_tab = torch.empty(max(len(flat), 2 * (tab.numel() if tab is not None else 0)), dtype=torch.int64, device=eng.device)
                h.chunk_tab, h.nchunks = tab, len(flat) // 2
                prog = Program(P.const_i64_descs(tab.data_ptr(), flat) + eng._optimizer_descs(h, Bg, cat_all, sg_all, clip, eps, rank_layout=rank_layout))
                prog.holder = h
                plan.union_opt = prog  # (kept alive until the plan goes)
                prog.run(eng._sp())
                return
            # weight-sharing supernet: zero_grad / norm / Adagrad cover the arena ranges of the plan's path only (engine.compile),
            # so the optimizer program is the plan's: same work buffers, the plan's chunk table
            if getattr(plan, "opt_prog", None) is None:
                h = _Holder()
                for k in shared:
                    if hasattr(holder, k):
                        setattr(h, k, getattr(holder, k))
                h.chunk_tab, h.nchunks = plan.cp.chunk_tab, plan.cp.nchunks
                plan.opt_prog = Program(eng._optimizer_descs(h, Bg, cat_all, sg_all, clip, eps, rank_layout=rank_layout))
                plan.opt_prog.holder = h
            plan.opt_prog.run(eng._sp())

        return run

    def dp_path_spans(self, choice, B, grad_scale, clip, eps):
        """arena ranges [(offset, numel)] of the dense parameters the path `choice` trains (its plan knows: engine.compile — the
        arguments are dp_plan's, so this rank's own choice hits the plan it is about to run)"""
        cp = self.engine.compile(choice, B, True, clip, eps, graph=False, grad_scale=grad_scale, defer_dw=False, local_optimizer=False)
        return list(cp.path_spans)

    def dp_dedup_ids(self):
        """launcher of the id-only half of the global batch's row dedup (None: this batch size runs the one-launch kernels): to be
        enqueued once the ids all-gather has landed, anywhere before the optimizer"""
        import ctypes as C
        from . import _lib as L
        d = getattr(self._holder, "dedup_ids", None)
        if d is None:
            return None
        eng, lib = self.engine, L.load()
        return lambda: L.check(lib.nasrec_launch(eng._sp(), C.addressof(d)))

    def dp_tail_ops(self, send, recv, rows_n, tail, world):
        """(pack, unpack) of the dense gradients that ride in the row-gradient all-gather, one launch each: COPY_SEGS of the arena ranges
        behind this rank's rows; REDUCE_ROWS over the W ranks' copies (fixed = rank order) back into the arena"""
        import ctypes as C
        from . import _lib as L
        eng, lib = self.engine, L.load()
        tail_n = sum(n for _, n in tail)
        stride = rows_n + tail_n
        if len(tail) > L.MAX_SEGS or stride >= (1 << 31):
            return _torch_tail_ops(eng.flat_g, send, recv, rows_n, tail, world)
        cs = L.CopySegsDesc()
        cs.kind, cs.B, cs.nseg, cs.ld_dst = L.OP_COPY_SEGS, 1, len(tail), tail_n
        cs.dst = send.data_ptr() + 4 * rows_n
        rr = L.ReduceRowsDesc()
        rr.kind, rr.R, rr.C, rr.ld = L.OP_REDUCE_ROWS, world, tail_n, stride
        rr.in_ = recv.data_ptr() + 4 * rows_n
        rr.ndst = len(tail)
        pos = 0
        for q, (off, n) in enumerate(tail):
            cs.seg[q], cs.width[q], cs.ld[q], cs.off[q] = eng.flat_g.data_ptr() + 4 * off, n, n, pos
            rr.dst[q], rr.dst_off[q], rr.dst_len[q] = eng.flat_g.data_ptr() + 4 * off, pos, n
            pos += n
        keep = (cs, rr)
        return (lambda: L.check(lib.nasrec_launch(eng._sp(), C.addressof(keep[0])))), (lambda: L.check(lib.nasrec_launch(eng._sp(), C.addressof(keep[1]))))

    def dp_plan(self, choice, B, grad_scale, clip, eps, graph, row_grad_out=None) -> DPPlan:
        from .engine import Program
        eng = self.engine
        fixed = eng.cfg.fixed
        # weight-gradient products stay in backward order (never parked behind the backward): a block's gradients are complete when
        # its backward is, and can travel under the blocks that follow
        # (local_optimizer = False: clip + Adagrad of a data-parallel step run over the GLOBAL batch, dp_optimizer)
        cp = eng.compile(choice, B, True, clip, eps, graph=False, grad_scale=grad_scale, defer_dw=False, row_grad_out=row_grad_out, local_optimizer=False)
        plan = DPPlan()
        plan.cp = cp
        plan.cat_local, plan.loss = cp.cat_x, cp.loss
        plan.sparse_grad = cp.sparse0.grad_tensor()
        plan.stage = lambda int_x, cat_x, y, lr: eng._stage_inputs(eng._sp(), cp, int_x, cat_x, y, lr)
        descs = cp.bwd.descs
        with torch.cuda.stream(eng.stream):
            if fixed:
                # forward + backward as ONE launch list (batch <= 256: the level-scheduled joint program, whose DAG places every
                # weight-gradient product as early as its operands allow), cut where enough finished gradient has piled up: each
                # piece is followed by the all-reduce of the arena ranges that are FINAL after it (no later launch writes them),
                # which then travels under the rest of the backward.  Parameters nothing writes (operators off the backward's
                # live set: grad None in the reference) stay zero on every rank and are not sent.
                descs = list(cp.fb.descs) if getattr(cp, "fb", None) is not None else list(cp.fwd.descs) + list(cp.bwd.descs)
                ready = gradient_ready_index(eng, descs)
                pieces = cut_segments(ready, len(descs), {n: eng.params[n].numel() for n in ready}, DP_SEGMENTS)
                plan.forward = lambda: None
                plan.ready, plan.cuts = ready, [end for end, _ in pieces]
                segs, start = [], 0
                order = sorted(eng.offsets, key=lambda n: eng.offsets[n])
                # a merged range runs over gradients NO launch of this step writes.  They must be zero on every rank for ever (an in-place
                # all-reduce multiplies a leftover by the world size every step, and the whole-arena sum of squares feeds it into the clip):
                # zeroed here, once — nothing of the plan writes them afterwards (`NASREC_DP_DEBUG=1` checks it after every eager step)
                plan.never_written = [n for n in order if n not in ready]
                for n in plan.never_written:
                    eng.grads[n].zero_()
                for end, names in pieces:
                    prog = Program(descs[start:end])
                    segs.append(((lambda p: (lambda: p.run(eng._sp())))(prog), merge_over_unwritten(eng, order, set(names), ready)))
                    start = end
                plan.segments = segs
            else:
                plan.forward = lambda: cp.fwd.run(eng._sp())
                marks = sorted(cp.bwd_marks, key=lambda m: m[1])  # (block, end index in the backward program), ascending position
                used = [n for n in list(cp.ctx.grad_params) + ["_final.weight", "_final.bias"] if not n.startswith("_embedding.")]
                groups, sent = [], set()
                for blk, end in marks:
                    names = [n for n in used if n not in sent and (n.startswith("_blocks.%d." % blk) or n.startswith("_final."))]
                    sent.update(names)
                    groups.append((end, names))
                groups.append((len(descs), [n for n in used if n not in sent]))
                span = lambda n: (eng.offsets[n], eng.params[n].numel())
                buckets = [coalesce_ranges([span(n) for n in names]) for _, names in groups]
                # A merged range also covers the gaps between its members.  (1) No parameter that ANOTHER bucket sends may sit in such
                # a gap (it would be reduced twice, or while its backward is still writing it).  (2) What does sit there are gradients
                # of parameters off this path: nothing of this step writes or reads them, but summing leftovers of earlier paths in
                # place, step after step, grows without bound — they are zeroed at the start of every backward.
                gaps = []
                for bi, ranges in enumerate(buckets):
                    own = sorted(span(n) for n in groups[bi][1])
                    others = [span(n) for bj, (_, names) in enumerate(groups) if bj != bi for n in names]
                    for off, n in ranges:
                        assert not any(o < off + n and off < o + m for o, m in others), "a gradient bucket's gap spans a parameter of another bucket"
                        pos = off
                        for o, m in own:
                            if off <= o < off + n:
                                if o > pos:
                                    gaps.append((pos, o - pos))
                                pos = max(pos, o + m)
                        if off + n > pos:
                            gaps.append((pos, off + n - pos))
                head = []
                if gaps:
                    from . import plan as P
                    flat = P.path_chunks(gaps)
                    plan.gap_tab = torch.empty(len(flat), dtype=torch.int64, device=eng.device)
                    ms = P.memset_desc(eng.flat_g)
                    ms.chunks, ms.nchunks = plan.gap_tab.data_ptr(), len(flat) // 2
                    head = P.const_i64_descs(plan.gap_tab.data_ptr(), flat) + [ms]
                segs, start = [], 0
                for (end, _), ranges in zip(groups, buckets):
                    prog = Program(head + descs[start:end])
                    head = []
                    segs.append(((lambda p: (lambda: p.run(eng._sp())))(prog), ranges))
                    start = end
                plan.segments = segs
        eng.stream.synchronize()
        return plan


class _Holder:
    pass


def _jsonable(o):
    if hasattr(o, "tolist"):
        return o.tolist()
    if hasattr(o, "item"):
        return o.item()
    raise TypeError(type(o))
