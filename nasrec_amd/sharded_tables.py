"""Row-sharded embedding tables: the rows of every table are split over the ranks, ids / rows / row gradients travel by
all-to-all (RCCL over xGMI on the GPUs, gloo in the CPU tests) — for tables that outgrow one GPU's HBM.  Bag size 1, as the reference's
stem (`supernet.py:404-410,418-428`: Fs independent nn.Embedding look-ups stacked on dim 1).

Placement: table f (n_f rows) is cut into `world` contiguous row ranges of rp_f = ceil(n_f / world) rows; rank r owns rows
[r rp_f, min(n_f, (r + 1) rp_f)).  Row-wise (not table-wise) sharding balances any mix of table sizes (Criteo: 10 M-row and 3-row
tables side by side).

One training step (the dense network stays data-parallel, `nasrec_amd/parallel.py`):
  1. route      every (sample b, field f) item goes to the owner of its row, in the slot [b, f] of the [B, Fs] matrix the owner receives
                from this rank (PAD_ID in the slots of items other ranks own): the owner's [W B, Fs] matrix is in GLOBAL batch order
                (source rank, then b) — the order in which duplicate rows are summed;
  2. lookup     all-to-all of the ids, owners gather their rows (bit-exact copy; zero rows for the pads), all-to-all of the rows back,
                every item picks its row out of its owner's slot: [B, Fs, 16] for the local forward;
  3. backward   the local row gradients [B, Fs, 16] take the same route to the owners (one all-to-all);
  4. update     every owner runs the row-sparse dedup + clip + Adagrad on ITS rows; the clip coefficient needs the global gradient norm:
                the owners' partial sums of squares are all-reduced (a few hundred bytes).
Equivalence target (tests/test_sharded_tables_cpu.py, gloo world 2): the step of a single process at the global batch with whole tables.

Routing (round 5): FIXED-CAPACITY slots — every rank sends every peer a full [B, Fs] slot with PAD_ID where the peer does not own the
row — so there is no sort, no compaction and no split size for the host to read back: a step enqueues without synchronising with the
device.  What runs on the rows is pluggable: the HIP engine's gather / dedup / Adagrad kernels on the GPU (`EngineShardedOps` in this
file binds them), index ops in the CPU tests."""
import math
import os as _os
from typing import List, Optional

import torch
import torch.distributed as dist

E = 16
PAD_ID = -(1 << 30)  # pad of the owner-side id matrix: out of every table's range (the row update skips it), unlike any sentinel of the dedup kernels


class Route:
    """how the items of one local batch travel: built by RowShardedTables.route(), used for the forward and the backward exchange"""
    __slots__ = ("B", "owner", "own_idx", "Bp")


class RowShardedTables:
    def __init__(self, num_embeddings: List[int], device, group=None, dtype=torch.float32, init_fn=None, shards=None):
        """num_embeddings: rows of every (whole) table.  init_fn(f, row_lo, row_hi) -> [row_hi - row_lo, 16] tensor initialises a
        shard (default: zeros); shards: this rank's row ranges as existing [max(rows, 1), 16] tensors, ADOPTED (no copy: the drop-in
        module hands over its nn.Embedding weights)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.num_embeddings = [int(n) for n in num_embeddings]
        self.Fs = len(self.num_embeddings)
        self.device = torch.device(device)
        self.rp = [max(1, math.ceil(n / self.world)) for n in self.num_embeddings]  # rows per rank of table f
        self.lo = [min(n, self.rank * rp) for n, rp in zip(self.num_embeddings, self.rp)]
        self.hi = [min(n, (self.rank + 1) * rp) for n, rp in zip(self.num_embeddings, self.rp)]
        self.tables, self.state = [], []
        for f in range(self.Fs):
            rows = max(self.hi[f] - self.lo[f], 1)  # (an empty shard keeps one unused row so that every pointer is valid)
            if shards is not None:
                t = shards[f]
                assert tuple(t.shape) == (rows, E) and t.is_contiguous() and t.device.type == self.device.type, (f, tuple(t.shape), rows)
                self.tables.append(t)
                self.state.append(torch.zeros_like(t))
                continue
            t = init_fn(f, self.lo[f], self.hi[f]).to(self.device, dtype) if (init_fn is not None and self.hi[f] > self.lo[f]) else \
                torch.zeros(rows, E, dtype=dtype, device=self.device)
            assert tuple(t.shape) == (rows, E)
            self.tables.append(t.contiguous())
            self.state.append(torch.zeros_like(self.tables[-1]))
        self._rp_t = torch.tensor(self.rp, dtype=torch.int64, device=self.device)
        self._n_t = torch.tensor(self.num_embeddings, dtype=torch.int64, device=self.device)
        self._dest = torch.arange(self.world, dtype=torch.int64, device=self.device).view(self.world, 1, 1)
        self.oob = torch.zeros((), dtype=torch.bool, device=self.device)  # an id outside its table was seen (check_indices)

    # ---------------------------------------------------------------------------------------------------------------------
    # Fixed-capacity exchange (round 5).  Every rank sends every peer a FULL [B, Fs] slot: the entries the peer owns carry the local
    # row number, the others PAD_ID (ids) / zeros (rows, gradients).  No split sizes, no sort, no compaction — hence nothing for the
    # host to read back: the whole step enqueues without a synchronisation — and the owner sees source s's sample b at row s B + b of
    # its [W B, Fs] matrix, i.e. in GLOBAL batch order (the order in which duplicate rows are summed).  The price is W x the bytes of a
    # compacted exchange (W B Fs 64 B per rank and direction: 3.4 MB at 8 x 256 x 26), which at these sizes is latency, not bandwidth.
    def route(self, cat_x: torch.Tensor, mask_fn=None) -> Route:
        """cat_x [B, Fs] int64 (global row ids) -> Route.  No host synchronisation.  mask_fn(cat_x) -> (owner [B, Fs] int64,
        send_ids [W, B, Fs] int64) runs the masking on the engine (one launch; default: tensor operations)."""
        B, W = int(cat_x.shape[0]), self.world
        r = Route()
        r.B, r.Bp = B, W * B
        if mask_fn is not None:
            r.owner, send_ids = mask_fn(cat_x)
        else:
            r.owner = torch.div(cat_x, self._rp_t.view(1, self.Fs), rounding_mode="floor")
            local = cat_x - r.owner * self._rp_t.view(1, self.Fs)
            self.oob |= ((cat_x < 0) | (cat_x >= self._n_t.view(1, self.Fs))).any()
            send_ids = torch.where(r.owner.unsqueeze(0) == self._dest, local.unsqueeze(0), torch.full_like(local, PAD_ID).unsqueeze(0))
        r.own_idx = self._a2a(send_ids.view(W, B * self.Fs)).view(W * B, self.Fs)  # row s B + b = source s, sample b
        return r

    def _a2a(self, send: torch.Tensor) -> torch.Tensor:
        """all_to_all_single with equal splits along dim 0 (one [W, ...] slot per peer)"""
        if self.world == 1:
            return send
        out = torch.empty_like(send)
        dist.all_to_all_single(out, send.contiguous(), group=self.group)
        return out

    # ---------------------------------------------------------------------------------------------------------------------
    def lookup(self, cat_x: torch.Tensor, gather_fn=None, mask_fn=None, select_fn=None):
        """-> (rows [B, Fs, 16] of the LOCAL batch, Route).  gather_fn(own_idx [W B, Fs], pads = PAD_ID) -> [W B, Fs, 16] with zero rows
        for the pads runs the owner-side gather on this rank's shards; select_fn(back [W, B, Fs, 16], owner) -> [B, Fs, 16] picks every
        item's row out of its owner's slot (defaults: tensor operations)."""
        r = self.route(cat_x, mask_fn)
        W, B, Fs = self.world, r.B, self.Fs
        if gather_fn is not None:
            own_rows = gather_fn(r.own_idx)
        else:
            keep = (r.own_idx >= 0).unsqueeze(-1)
            safe = r.own_idx.clamp_min(0)
            own_rows = torch.stack([self.tables[f][safe[:, f].clamp_max(self.tables[f].shape[0] - 1)] for f in range(Fs)], 1) * keep
        back = self._a2a(own_rows.reshape(W, B * Fs * E)).view(W, B, Fs, E)  # slot o = what owner o holds of my batch
        if select_fn is not None:
            rows = select_fn(back, r.owner)
        else:
            rows = torch.gather(back, 0, r.owner.clamp(0, W - 1).view(1, B, Fs, 1).expand(1, B, Fs, E)).squeeze(0)
        return rows.contiguous(), r

    def send_grads(self, r: Route, sg: torch.Tensor, mask_fn=None):
        """local row gradients [B, Fs, 16] -> owner-side (own_idx [W B, Fs] with pads PAD_ID, grads [W B, Fs, 16] with zero pads)"""
        W, B, Fs = self.world, r.B, self.Fs
        sg = sg.view(B, Fs, E)
        if W == 1:
            return r.own_idx, sg
        if mask_fn is not None:
            send = mask_fn(sg, r.owner)
        else:
            send = torch.where((r.owner.unsqueeze(0) == self._dest).unsqueeze(-1), sg.unsqueeze(0), torch.zeros((), dtype=sg.dtype, device=sg.device))
        got = self._a2a(send.reshape(W, B * Fs * E))
        return r.own_idx, got.view(W * B, Fs, E)

    def check_indices(self):
        """raise IndexError if any id seen so far was outside its table (torch raises at lookup time; here the flag is read on demand:
        the step itself never waits for the device)"""
        if bool(self.oob):
            raise IndexError("index out of range in embedding lookup (row-sharded tables)")

    # ---------------------------------------------------------------------------------------------------------------------
    def reference_update(self, own_idx, own_g, coef: float, lr: float, eps: float):
        """the owner-side row-sparse Adagrad in plain torch (CPU tests; the GPU path runs the HIP dedup + Adagrad kernels): duplicates
        of a row are summed in arrival (= global batch) order, then g' = g coef; state += g'^2; p -= lr g' / (sqrt(state) + eps)"""
        for f in range(self.Fs):
            ids = own_idx[:, f]
            keep = ids >= 0
            if not bool(keep.any()):
                continue
            uniq, inv = torch.unique(ids[keep], return_inverse=True)
            g = torch.zeros(uniq.numel(), E, dtype=own_g.dtype, device=own_g.device).index_add_(0, inv, own_g[keep, f]) * coef
            self.state[f][uniq] += g * g
            self.tables[f][uniq] -= lr * g / (self.state[f][uniq].sqrt() + eps)

    def grad_sumsq(self, own_idx, own_g) -> torch.Tensor:
        """this rank's share of the squared gradient norm of the tables (duplicates summed first)"""
        tot = torch.zeros((), dtype=torch.float64, device=own_g.device)
        for f in range(self.Fs):
            ids = own_idx[:, f]
            keep = ids >= 0
            if not bool(keep.any()):
                continue
            uniq, inv = torch.unique(ids[keep], return_inverse=True)
            g = torch.zeros(uniq.numel(), E, dtype=torch.float64, device=own_g.device).index_add_(0, inv, own_g[keep, f].double())
            tot += (g * g).sum()
        return tot

    def whole_table(self, f: int, which: str = "tables") -> torch.Tensor:
        """all-gather of table f (which = "tables"), or of its Adagrad accumulator ("state") — tests / checkpoints; a collective"""
        src = self.tables if which == "tables" else self.state
        n, rp = self.num_embeddings[f], self.rp[f]
        mine = torch.zeros(rp, E, dtype=src[f].dtype, device=self.device)
        k = self.hi[f] - self.lo[f]
        if k > 0:
            mine[:k] = src[f][:k]
        if self.world > 1:
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
        else:
            parts = [mine]
        return torch.cat(parts, 0)[:n]


# ----------------------------------------------------------------------------------------------------------------
# the training step over row-sharded tables
# ----------------------------------------------------------------------------------------------------------------
class ShardedTableStep:
    """One training step with row-sharded tables and a data-parallel dense network, over a small protocol (so that the gloo tests run
    THIS code with the oracle behind it, and the GPUs run it with the HIP engine behind it — EngineShardedOps below):
        ops.forward_backward(int_x, rows [B,Fs,16], y, choice, grad_scale) -> (loss, row gradients [B,Fs,16]); dense gradients in ops.flat_g
        ops.dense_sumsq() -> 0-d float64 tensor (over the step's path only);  ops.dense_update(coef 0-d tensor, lr)
        ops.grad_ranges() -> [(offset, numel)] of ops.flat_g that this step's path wrote, or None for the whole arena (optional)
        ops.gather(tables, own_idx) -> [W B, Fs, 16], zero rows where own_idx is PAD_ID (optional; default torch indexing)
        ops.rows_sumsq(tables, own_idx, own_g) -> 0-d float64;  ops.rows_update(tables, own_idx, own_g, coef, lr, eps)"""

    def __init__(self, ops, tables: RowShardedTables, B_local: int, clip: Optional[float] = 5.0, eps: float = 1e-2, graph: bool = True):
        """graph: when the ops say the step is capturable (`ops.capturable`: the HIP engine on a fixed sub-network — one plan, static
        buffers), the WHOLE step — routing, the three all-to-alls, forward / backward, the all-reduces, owner-side dedup + clip + Adagrad —
        is captured once (the second call) into one graph and replayed: a step is four small input copies and one graph launch."""
        self.ops, self.tables, self.B, self.clip, self.eps = ops, tables, B_local, clip, eps
        self.world = tables.world
        self.last_norm = None
        self.graph = bool(graph) and _os.environ.get("NASREC_SHARDED_GRAPH", "1") != "0"  # (env: A/B knob)
        self._g = None  # (key, CUDAGraph, static inputs, loss, the choice object it was captured for)
        self._calls = 0

    def step(self, int_x, cat_x, y, lr, choice=None):
        dev = self.tables.device
        if not (self.graph and dev.type == "cuda" and getattr(self.ops, "capturable", False)):
            return self._step(int_x, cat_x, y, lr, choice)
        self._calls += 1
        if self._calls == 1:  # the first step runs eagerly: plans are compiled, communicators created, work buffers allocated
            return self._step(int_x, cat_x, y, lr, choice)
        # keyed on the CONTENT of the choice (a caller that rebuilds an equal dict every step must not re-capture every step, and a
        # recycled id() after a collection must not replay another choice's graph); `is` = the fast path of an unchanged object
        if self._g is not None and self._g[4] is choice and self._g[0][1:] == (tuple(int_x.shape), tuple(cat_x.shape), int_x.dtype):
            key = self._g[0]
        else:
            import json
            from .engine import _jsonable
            key = (json.dumps(choice, sort_keys=True, default=_jsonable), tuple(int_x.shape), tuple(cat_x.shape), int_x.dtype)
        if self._g is None or self._g[0] != key:
            static = (torch.empty_like(int_x), torch.empty_like(cat_x), torch.empty_like(y), torch.zeros(1, dtype=torch.float32, device=dev))
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            ok, loss = True, None
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    loss = self._step(static[0], static[1], static[2], static[3], choice)
            except Exception as e:  # noqa: BLE001
                import warnings
                warnings.warn("row-sharded step: graph capture failed (%s); running eagerly" % (e,))
                torch.cuda.synchronize(dev)
                ok = False
            if self.world > 1:
                # every rank replays or every rank runs eagerly: a rank that fell back alone would issue its collectives from the host
                # while the others issue theirs from a graph replay — the same sequence, but a capture that failed HALFWAY has already
                # left the ranks' communicators out of step; agree on the outcome (MIN) before anybody chooses
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.tables.group)
                ok = bool(int(flag.item()))
            if not ok:
                self.graph = False
                return self._step(int_x, cat_x, y, lr, choice)
            self._g = (key, g, static, loss, choice)
        _, g, static, loss, _ = self._g
        static[0].copy_(int_x, non_blocking=True)
        static[1].copy_(cat_x, non_blocking=True)
        static[2].copy_(y.reshape(static[2].shape), non_blocking=True)
        static[3].fill_(float(lr))
        g.replay()
        return loss

    def _step(self, int_x, cat_x, y, lr, choice=None):
        t, ops = self.tables, self.ops
        gather = getattr(ops, "gather", None)
        rows, route = t.lookup(cat_x, (lambda idx: gather(t, idx)) if gather is not None else None)
        loss, sg = ops.forward_backward(int_x, rows, y, choice, 1.0 / (self.B * self.world))
        if self.world > 1:
            # dense gradients: sum over the ranks (1 / (B world) is folded into dlogits).  A sampled supernet path writes (and zeroes)
            # only ITS ranges of the gradient arena; what lies outside are leftovers of earlier paths, which nothing reads — summing
            # them in place over the ranks, step after step, would grow them by a factor `world` per step until they overflow, and
            # the clip norm must not see them either: only the path's ranges travel and count (ops.grad_ranges; None = whole arena)
            ranges = ops.grad_ranges() if hasattr(ops, "grad_ranges") else None
            if ranges is None:
                dist.all_reduce(ops.flat_g, group=t.group)
            else:
                for off, n in ranges:
                    dist.all_reduce(ops.flat_g[off:off + n], group=t.group)
        own_idx, own_g = t.send_grads(route, sg)
        ss = ops.rows_sumsq(t, own_idx, own_g) if hasattr(ops, "rows_sumsq") else t.grad_sumsq(own_idx, own_g)
        if self.world > 1:
            dist.all_reduce(ss, group=t.group)             # every owner's share of the tables' squared gradient norm
        # the squares are summed in fp64; norm and coefficient then in fp32, operation for operation what torch.nn.utils.clip_grad_norm_
        # (train_utils.py:285) and the whole-table step's kernel (csrc/optimizer_bodies.h clip_coef_wave) compute: total = float(sqrt(s));
        # coef = min(max_norm / (total + 1e-6f), 1) — in fp64 throughout, the coefficient could differ from theirs by one ulp
        # (ops.norm_dtype: fp32 for the HIP engine; the fp64 oracle behind the same protocol keeps fp64 throughout)
        nd = getattr(ops, "norm_dtype", torch.float64)
        total = torch.sqrt(ops.dense_sumsq() + ss).to(nd)
        if self.clip is not None:
            k = (nd, total.device)
            if getattr(self, "_consts", (None,))[0] != k:  # (built once: a host scalar turned into a device tensor per step is a copy per step)
                self._consts = (k, torch.tensor(self.clip, dtype=nd, device=total.device), torch.tensor(1e-6, dtype=nd, device=total.device))
            coef = torch.clamp(self._consts[1] / (total + self._consts[2]), max=1.0)
        else:
            coef = torch.ones_like(total)
        self.last_norm = total
        ops.dense_update(coef, lr)
        if hasattr(ops, "rows_update"):
            ops.rows_update(t, own_idx, own_g, coef, lr, self.eps)
        else:
            t.reference_update(own_idx, own_g, float(coef), lr, self.eps)
        return loss


class EngineShardedOps:
    """The protocol of ShardedTableStep on the HIP engine: a SupernetEngine in `host_embedding` mode (it holds no table; the looked-up
    rows come with the batch) for the network, and the engine's gather / dedup / row-Adagrad kernels on this rank's table shards."""

    def __init__(self, engine, clip=5.0, eps=1e-2):
        import ctypes as C
        from . import _lib as L
        assert engine.host_embedding, "build the engine with host_embedding=True: the tables live in RowShardedTables"
        self.eng, self.eps, self.L, self.C = engine, eps, L, C
        self.flat_g = engine.flat_g
        self.coef_dev = torch.ones(2, dtype=torch.float32, device=engine.device)
        self.norm_dtype = torch.float32  # the clip coefficient in the arithmetic of the whole-table step's kernel (ShardedTableStep.step)
        self._bufs = {}
        self.capturable = bool(engine.cfg.fixed)  # one plan, static buffers: the whole step can be captured as one graph
        self._zero_ids = None

    def _launch(self, desc):
        self.L.check(self.L.load().nasrec_launch(self.eng._sp(), self.C.addressof(desc)))

    def forward_backward(self, int_x, rows, y, choice, grad_scale):
        eng = self.eng
        choice = choice if choice is not None else eng.warm_choice
        cp = eng.compile(choice, int(int_x.shape[0]), train=True, grad_scale=grad_scale)
        sp = eng._sp()
        if self._zero_ids is None or self._zero_ids.shape[0] != int_x.shape[0]:
            self._zero_ids = torch.zeros(int_x.shape[0], eng.Fs, dtype=torch.int64, device=eng.device)
        eng._stage_inputs(sp, cp, int_x, self._zero_ids, y.reshape(-1), rows=rows)
        (cp.fb if getattr(cp, "fb", None) is not None else cp.fwd).run(sp)
        if getattr(cp, "fb", None) is None:
            cp.bwd.run(sp)
        self.cp = cp
        return cp.loss, cp.sparse0.grad_tensor().view(int_x.shape[0], eng.Fs, E)

    def grad_ranges(self):
        """arena ranges of the parameters the last step's path trained (engine.compile: `cp.path_spans`, merged over alignment padding
        only); None for a fixed sub-network, whose whole arena is written or zero"""
        spans = getattr(self.cp, "path_spans", None)
        if spans is None:
            return None
        from .parallel import coalesce_ranges
        return coalesce_ranges(spans, gap=3)

    def dense_sumsq(self):
        """squared norm of the dense gradients of the step's path: the engine's own fixed-order square-sum launch over the path's
        chunk table (what `_optimizer_descs` runs in the plain step), partial sums added in fp64"""
        L, eng = self.L, self.eng
        tab, ntab = getattr(self.cp, "chunk_tab", None), getattr(self.cp, "nchunks", 0)
        if getattr(self, "_sq_partial", None) is None:
            self._sq_partial = torch.zeros(256, dtype=torch.float32, device=eng.device)
        sq = L.SumsqDesc()
        sq.kind = L.OP_SUMSQ
        sq.nblocks = max(1, min(256, ntab if tab is not None else (eng.flat_numel + 256 * 8 - 1) // (256 * 8)))
        sq.n, sq.x, sq.partial = eng.flat_numel, eng.flat_g.data_ptr(), self._sq_partial.data_ptr()
        if tab is not None:
            sq.chunks, sq.nchunks = tab.data_ptr(), ntab
        self._launch(sq)
        return self._sq_partial[:sq.nblocks].double().sum()

    def dense_update(self, coef, lr):
        L, eng = self.L, self.eng
        self.coef_dev[0] = coef.to(torch.float32)
        if torch.is_tensor(lr):
            eng.lr_dev.copy_(lr.to(torch.float32).reshape(1))  # (a device scalar: the captured step is replayed with a new learning rate)
        else:
            eng.lr_dev.fill_(float(lr))
        ad = L.AdagradDenseDesc()
        ad.kind, ad.eps, ad.n = L.OP_ADAGRAD_DENSE, self.eps, eng.flat_numel
        ad.p, ad.g, ad.state = eng.flat_p.data_ptr(), eng.flat_g.data_ptr(), eng.flat_s.data_ptr()
        ad.lr, ad.coef = eng.lr_dev.data_ptr(), self.coef_dev.data_ptr()
        tab, ntab = getattr(self.cp, "chunk_tab", None), getattr(self.cp, "nchunks", 0)
        if tab is not None:
            ad.chunks, ad.nchunks = tab.data_ptr(), ntab
        self._launch(ad)

    def gather(self, t: RowShardedTables, idx_safe):
        L = self.L
        Bp = int(idx_safe.shape[0])
        out = torch.empty(Bp, t.Fs, E, dtype=torch.float32, device=idx_safe.device)
        g = L.EmbedDesc()
        g.kind, g.B, g.Fs = L.OP_EMBED_GATHER, Bp, t.Fs
        g.idx, g.out, g.oob = idx_safe.data_ptr(), out.data_ptr(), None  # (pads are PAD_ID: zero rows, no out-of-range flag — RowShardedTables.oob has the real ones)
        for f in range(t.Fs):
            g.table[f], g.rows[f] = t.tables[f].data_ptr(), t.tables[f].shape[0]
        self._launch(g)
        self._keep = (idx_safe, out)
        return out

    def _dedup(self, t, own_idx, own_g):
        L = self.L
        Bp = int(own_idx.shape[0])
        key = (Bp, t.Fs)
        if self._bufs.get("key") != key:
            dev = own_g.device
            nb = (Bp + 255) // 256
            self._bufs = dict(key=key, leader=torch.zeros(Bp * t.Fs, dtype=torch.int32, device=dev), gsum=torch.zeros(Bp * t.Fs * E, device=dev),
                              partial=torch.zeros(t.Fs * nb, device=dev))
        b = self._bufs
        dd = L.EmbDedupDesc()
        dd.kind, dd.B, dd.Fs = L.OP_EMB_DEDUP, Bp, t.Fs
        dd.idx, dd.dout = own_idx.data_ptr(), own_g.data_ptr()
        dd.leader, dd.gsum, dd.sumsq_partial = b["leader"].data_ptr(), b["gsum"].data_ptr(), b["partial"].data_ptr()
        dd.overflow = self.eng.oob.data_ptr() + 4
        self._launch(dd)
        return b

    def rows_sumsq(self, t, own_idx, own_g):
        self._own = (own_idx.contiguous(), own_g.contiguous())
        b = self._dedup(t, *self._own)
        return b["partial"].double().sum()

    def rows_update(self, t, own_idx, own_g, coef, lr, eps):
        L = self.L
        b = self._bufs
        own_idx, own_g = self._own
        ar = L.AdagradRowsDesc()
        ar.kind, ar.B, ar.Fs, ar.eps = L.OP_ADAGRAD_ROWS, int(own_idx.shape[0]), t.Fs, eps
        ar.idx, ar.leader, ar.gsum = own_idx.data_ptr(), b["leader"].data_ptr(), b["gsum"].data_ptr()
        for f in range(t.Fs):
            ar.table[f], ar.state[f], ar.rows[f] = t.tables[f].data_ptr(), t.state[f].data_ptr(), t.tables[f].shape[0]
        ar.lr, ar.coef = self.eng.lr_dev.data_ptr(), self.coef_dev.data_ptr()
        self._launch(ar)
