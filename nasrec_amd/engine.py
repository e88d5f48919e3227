"""SupernetEngine — executes compiled plans (nasrec_amd/plan.py) on one MI355X through the C-ABI.

Ownership: PyTorch owns every byte of device memory (parameters, gradients, optimizer state, activations);
the engine owns launch plans, a HIP stream and captured hipGraphs.  Dense parameters live in ONE flat fp32 buffer
(gradients and Adagrad state in two more of the same layout) so that the global-norm clip, the optimizer and the
data-parallel all-reduce are single passes; the embedding tables stay separate [rows,16] tensors and are updated
row-sparsely (identical to the dense reference update when weight_decay == 0).

There is no CPU path here: everything below needs the HIP library and a GPU.
"""
import ctypes as C
import json
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import plan as P
from . import schedule as S

E = 16
# largest (global) batch whose row-gradient dedup runs in two halves (csrc/dedup_bodies.h; <= L.DEDUP_IDS_MAX_B); 0 = the one-launch kernels
DEDUP_SPLIT_MAX_B = min(int(os.environ.get("NASREC_DEDUP_SPLIT_MAX_B", "2048")), L.DEDUP_IDS_MAX_B)
# the id half of a level-scheduled step (B <= 256) as an ITEM of the joint forward + backward program (A/B knob: 0 = on the staging launch)
_IDS_AS_ITEM = os.environ.get("NASREC_IDS_AS_ITEM", "1") != "0"
# NASREC_WL_TUNE=1: time a handful of level schedules of a fixed batch-256 training plan at compile time (_tune_levels).  Off by default: on the
# round-6 bodies the model's own schedule is within 0.4 % of the best candidate (0.2172 against 0.2163 ms per joint program), and the replays
# write the engine-wide gradient arena — a compile() in the middle of a step sequence must not do that.
_WL_TUNE = os.environ.get("NASREC_WL_TUNE", "0") == "1"
_WL_TUNE_MARGIN = float(os.environ.get("NASREC_WL_TUNE_MARGIN", "0.004"))
# the worklist descriptors of a fixed batch-256 plan live in device memory, uploaded once (ABI 17, nasrec_worklist_prepare): a launch that passes
# its 4 KB descriptor by value reads it from a fresh, cold copy in the kernel-argument ring (tools/micro/kernarg_probe.hip: + 0.75 - 1.6 us)
_WL_RESIDENT = os.environ.get("NASREC_WL_RESIDENT", "1") != "0"
_FUSE_FINAL = os.environ.get("NASREC_FUSE_FINAL", "1") != "0"  # joint program: final logit + the per-sample part of its backward as one operator
# clip + Adagrad of a fixed sub-network over the ranges of the parameters its backward reaches, not the whole arena (A/B knob)
_FIXED_OPT_TABLE = os.environ.get("NASREC_FIXED_OPT_TABLE", "1") != "0"


def _ptr_array(descs):
    arr = (C.c_void_p * len(descs))()
    for i, d in enumerate(descs):
        arr[i] = C.addressof(getattr(d, "launch_as", d))  # (a worklist whose descriptor lives in device memory: _resident_worklists)
    return arr


class Program:
    """An ordered list of descriptors; runs with one host call (nasrec_program_run) or as a captured hipGraph."""

    def __init__(self, descs: List):
        self.descs = list(descs)
        self.arr = _ptr_array(self.descs)
        self.n = len(self.descs)
        self.graph = None

    def run(self, stream_ptr):
        lib = L.load()
        L.check(lib.nasrec_program_run(stream_ptr, self.arr, self.n))

    def capture(self, stream_ptr):
        lib = L.load()
        g = C.c_void_p()
        L.check(lib.nasrec_graph_create(stream_ptr, self.arr, self.n, C.byref(g)))
        self.graph = g

    def replay(self, stream_ptr):
        L.check(L.load().nasrec_graph_launch(self.graph, stream_ptr))

    def __del__(self):
        try:
            if self.graph is not None:
                L.load().nasrec_graph_destroy(self.graph)
        except Exception:
            pass


class CompiledPlan:
    pass


_ITEMSIZE = {torch.float32: 4, torch.int64: 8, torch.int32: 4, torch.uint8: 1, torch.float64: 8}


class ArenaRef:
    """A buffer of a plan slot's arena: address and size now, the torch view only if somebody asks for one.  A sampled supernet path
    compiles ~2000 buffers per step and the plan only ever needs their ADDRESSES (descriptors carry raw pointers); creating a
    tensor view for each cost 2.5 ms of host time per step (40 % of the plan compile).  Anything else (`.view`, `.copy_`, `.double`,
    ...) goes to the lazily created tensor."""
    __slots__ = ("_chunk", "_off", "_n", "_dtype", "_ptr", "_t")

    def __init__(self, chunk, off, n, dtype, ptr):
        self._chunk, self._off, self._n, self._dtype, self._ptr, self._t = chunk, off, n, dtype, ptr, None

    def data_ptr(self):
        return self._ptr

    def numel(self):
        return self._n

    def tensor(self):
        if self._t is None:
            self._t = self._chunk[self._off:self._off + self._n * _ITEMSIZE[self._dtype]].view(self._dtype)
        return self._t

    def __getattr__(self, name):
        return getattr(self.tensor(), name)


class _RawDeviceBytes:
    """n bytes of device memory at ptr, as the CUDA array interface torch.as_tensor understands (zero-copy; the tensor's storage keeps this
    object alive, and the memory is freed with it)."""

    def __init__(self, ptr, n):
        self.ptr, self.n = int(ptr), int(n)
        self.pid = os.getpid()  # (a forked child — multiprocessing.Manager() of a test, a DataLoader worker — inherits this object but not a usable HIP
        #                         runtime: its garbage collector must not free the parent's memory.  Found as a segfault of a Manager server.)
        self.__cuda_array_interface__ = {"shape": (self.n,), "typestr": "|u1", "data": (self.ptr, False), "version": 2, "strides": None}

    def __del__(self):  # (when the last tensor view of the memory is gone: torch keeps this object alive through the storage)
        try:
            import sys
            if self.ptr and not sys.is_finalizing() and os.getpid() == self.pid:  # (at interpreter exit the HIP runtime may already be gone: the driver reclaims the memory)
                L.load().nasrec_free_uncached(C.c_void_p(self.ptr))
            self.ptr = 0
        except Exception:
            pass


class Arena:
    """Device memory of one plan slot: buffers are bump-allocated out of a few large chunks that live as long as the engine, so
    compiling a plan costs no allocator calls (a sampled supernet path almost never repeats: every step compiles a plan with
    ~10^3 buffers of new sizes, which sent the caching allocator to hipMalloc / hipFree and their device synchronisations).
    Slots are recycled in stream order: a plan's kernels are enqueued behind those of the plan that used the slot before."""

    CHUNK = 256 << 20  # bytes

    def __init__(self, device, uncached: bool = False):
        """uncached: the chunks are UNCACHED device memory (nasrec_alloc_uncached; wrapped as torch tensors through the CUDA array
        interface, freed with the arena): the arena of a persistent step, whose items hand buffers to each other inside one launch"""
        self.device = device
        self.uncached = bool(uncached)
        self.chunks: List[torch.Tensor] = []
        self.base: List[int] = []
        self.cur, self.off = 0, 0

    def reset(self):
        self.cur, self.off = 0, 0

    def _grow(self, nbytes):
        n = max(self.CHUNK, nbytes) if not self.uncached else max(64 << 20, (nbytes + (1 << 20) - 1) & ~((1 << 20) - 1))
        if self.uncached:
            with torch.cuda.device(self.device):
                ptr = C.c_void_p()
                L.check(L.load().nasrec_alloc_uncached(n, C.byref(ptr)))
            c = torch.as_tensor(_RawDeviceBytes(ptr.value, n), device=self.device)
            assert c.data_ptr() == ptr.value and c.numel() == n and c.dtype == torch.uint8
        else:
            c = torch.empty(n, dtype=torch.uint8, device=self.device)
        self.chunks.append(c)
        self.base.append(c.data_ptr())

    def ranges(self):
        """[(first byte, one past the last)] of the arena's chunks"""
        return [(b, b + c.numel()) for b, c in zip(self.base, self.chunks)]


    def alloc(self, numel: int, dtype=torch.float32) -> ArenaRef:
        numel = int(numel)
        nbytes = (numel * _ITEMSIZE[dtype] + 255) & ~255
        while True:
            if self.cur < len(self.chunks):
                c = self.chunks[self.cur]
                if self.off + nbytes <= c.numel():
                    r = ArenaRef(c, self.off, numel, dtype, self.base[self.cur] + self.off)
                    self.off += nbytes
                    return r
                self.cur += 1
                self.off = 0
                continue
            self._grow(nbytes)

    def used_bytes(self) -> int:
        return sum(c.numel() for c in self.chunks[:self.cur]) + self.off

    def reserve(self, nbytes: int):
        """grow to at least nbytes now (one allocator call per chunk, outside any timed or latency-sensitive region)"""
        while sum(c.numel() for c in self.chunks) < nbytes:
            self._grow(self.CHUNK)


# The buffers of a level-scheduled training plan (fixed sub-network, batch <= 256) live in UNCACHED device memory: a step of ~27 dependent
# launches hands every activation from one kernel to the next exactly once, and a store that goes through to memory leaves nothing for the
# kernel boundary to write back — 0.2485 -> 0.2447 / 0.2458 ms per cfg-2 step in four A/B pairs (profiles/r06_ab_uc_arena.txt), bit-identical
# results.  It is also what lets the persistent step (NASREC_OP_PERSIST) hand buffers over INSIDE a launch.  NASREC_UC_ARENA=0: torch's allocator.
_UC_ARENA = os.environ.get("NASREC_UC_ARENA", "1") == "1"
_UC_FLAT = os.environ.get("NASREC_UC_FLAT", "g")


def _on_device(fn):
    """run an engine entry point with the engine's device current: raw kernel launches, memsets, graph capture and event
    creation all act on the CURRENT device, whatever device the caller left selected (main_train.py --gpu N, N != 0)"""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        if torch.cuda.current_device() == self.device.index:
            return fn(self, *a, **kw)
        with torch.cuda.device(self.device):
            return fn(self, *a, **kw)
    return wrapped


class SupernetEngine:
    def __init__(self, cfg: P.NetConfig, Fd: int, Fs: int, num_embeddings: List[int], device="cuda:0", warm_choice=None,
                 world_size: int = 1, tables: Optional[List[torch.Tensor]] = None, host_embedding: bool = False):
        """host_embedding: the tables live in host memory (SuperNet(place_embedding_on_cpu=True), supernet.py:231,418-428): the engine
        holds no table, the caller hands the looked-up rows [B, Fs, 16] to every forward and receives their gradient; only the
        forward / backward programs are available in this mode (the fused optimizer step needs the tables on the device)."""
        L.load()
        if not torch.cuda.is_available():
            raise L.EngineError("SupernetEngine needs a GPU: there is no CPU fallback")
        self.cfg, self.Fd, self.Fs = cfg, Fd, Fs
        self.num_embeddings = [int(n) for n in num_embeddings[:Fs]]
        # hard limits of the kernels, raised here instead of corrupting later: the row-gradient dedup packs ids into 32 bits
        # (csrc/embedding.hip; ids are range-checked against the table by the gather, so rows < 2^31 makes the cast exact), a
        # descriptor carries at most MAX_TABLES table pointers
        for f, n in enumerate(self.num_embeddings):
            if not 0 < n < (1 << 31):
                raise ValueError("embedding table %d has %d rows: the engine supports 1 .. 2^31 - 1 rows per table" % (f, n))
        if Fs > L.MAX_TABLES:
            raise ValueError("%d sparse fields: the engine supports at most %d" % (Fs, L.MAX_TABLES))
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.world_size = world_size
        self.stream = torch.cuda.Stream(device=self.device)
        # the persistent step (NASREC_OP_PERSIST: the joint program's items in one launch, dependencies resolved in the kernel): built, bit-identical
        # to the level launches, and SLOWER on the cfg-2 step (0.274 against 0.245 ms: workgroups are dispatched in index order, and the ones
        # waiting for a dependency hold the chip's 768 slots against items that could run — DESIGN.md, profiles/r06_persist_*): opt-in
        self.persist = S.PERSIST and os.environ.get("NASREC_PERSIST_DEFAULT", "0") == "1"
        self._persist_errs = []
        self._last_plan = None
        # batch <= 256 (fixed sub-networks): operators that do not depend on each other share heterogeneous launches
        # (nasrec_amd/schedule.py); NASREC_WORKLIST=0 keeps one launch per operator (A/B runs, bit-identical results)
        self.level_schedule = os.environ.get("NASREC_WORKLIST", "1") != "0"
        # opt-in: drop forward operators nobody reads (schedule.eliminate_dead_forward); off = every reference operator runs
        self.dead_code_elimination = os.environ.get("NASREC_DCE", "0") == "1"
        if cfg.fixed:
            assert warm_choice is not None, "fixed mode needs the fixed choice"
            self.warm_choice = warm_choice
        else:
            self.warm_choice = P.full_path_choice(cfg)
        self.shapes = P.infer_param_shapes(cfg, self.warm_choice, Fd, Fs, self.num_embeddings)
        self.dense_names = [n for n in self.shapes if not n.startswith("_embedding.")]
        # flat dense arena, every parameter 16-byte aligned
        self.offsets, off = {}, 0
        for n in self.dense_names:
            self.offsets[n] = off
            numel = 1
            for s in self.shapes[n]:
                numel *= s
            off += (numel + 3) // 4 * 4
        self.flat_numel = off
        with torch.cuda.stream(self.stream):
            # A fixed sub-network's gradient arena lives in uncached memory like its plan's buffers (_UC_ARENA): every gradient is written once
            # by the backward and read once by the optimizer's launches — 0.2447 -> 0.2426 ms per cfg-2 step, three A/B pairs
            # (profiles/r06_ab_uc_arena.txt).  NASREC_UC_FLAT = letters of the flat arenas to place there: g(radients), s(tate), p(arameters).
            uc = _UC_FLAT if (cfg.fixed and off * 4 <= (64 << 20) and _UC_ARENA) else ""
            self._uc_flat_arena = Arena(self.device, uncached=True) if uc else None

            def flat(tag):
                if tag in uc:
                    t = self._uc_flat_arena.alloc(off).tensor()
                    t.zero_()
                    return t
                return torch.zeros(off, dtype=torch.float32, device=self.device)
            self.flat_p, self.flat_g, self.flat_s = flat("p"), flat("g"), flat("s")
            self.host_embedding = bool(host_embedding)
            if self.host_embedding:
                self.tables = []
            elif tables is not None:  # adopt the caller's nn.Embedding storage (no copy; PyTorch keeps ownership)
                for t, n in zip(tables, self.num_embeddings):
                    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (n, E)
                self.tables = list(tables)
            else:
                self.tables = [torch.zeros(n, E, dtype=torch.float32, device=self.device) for n in self.num_embeddings]
            self.table_state: Optional[List[torch.Tensor]] = None
            self.lr_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
            self.clip_out = torch.ones(2, dtype=torch.float32, device=self.device)  # [coef, total_norm]
            self.oob = torch.zeros(2, dtype=torch.int32, device=self.device)  # [index out of range, dedup hash partition overflow]
        self.params: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        self.state: Dict[str, torch.Tensor] = {}
        for n in self.dense_names:
            shp = self.shapes[n]
            numel = 1
            for s in shp:
                numel *= s
            o = self.offsets[n]
            self.params[n] = self.flat_p[o:o + numel].view(shp)
            self.grads[n] = self.flat_g[o:o + numel].view(shp)
            self.state[n] = self.flat_s[o:o + numel].view(shp)
        for f in range(len(self.tables)):
            self.params["_embedding.%d.weight" % f] = self.tables[f]
        self._plans: Dict[str, CompiledPlan] = {}
        self._pcache: Dict[str, tuple] = {}
        self._spare_arenas: List[Arena] = []
        self.stream.synchronize()

    # -------------------------------------------------------------------------------------------------------
    @_on_device
    def load_params(self, src: Dict[str, torch.Tensor]):
        """copy values (any device/dtype) into the engine's storage; keys = reference state_dict names"""
        with torch.cuda.stream(self.stream):
            for k, v in src.items():
                if k not in self.params:
                    raise KeyError("unexpected parameter %s" % k)
                self.params[k].copy_(torch.as_tensor(v).to(torch.float32).reshape(self.params[k].shape))
        missing = [k for k in self.params if k not in src]
        self.stream.synchronize()
        return missing

    @_on_device
    def init_weights(self, seed: int = 0):
        """train_utils.py:70-89 applied to this parameter set: xavier_normal_ tables, xavier_uniform_ 2-D weights
        (nn.Linear and the MultiheadAttention projections alike), zero biases; LayerNorm stays at its constructor
        value (1, or LN_INIT = 0.17 for the two Transformer norms, modules.py:598,636-640) with zero bias."""
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        with torch.cuda.stream(self.stream):
            for name, p in self.params.items():
                is_ln = ("_ln" in name) or ("layernorm" in name)
                if name.startswith("_embedding."):
                    std = (2.0 / (p.shape[0] + p.shape[1])) ** 0.5
                    p.normal_(0.0, std, generator=g)
                elif is_ln:
                    if name.endswith(".weight"):
                        p.fill_(0.17 if ("_attn_ln" in name or "_attn_fc_ln" in name) else 1.0)
                    else:
                        p.zero_()
                elif p.dim() == 2:
                    bound = (6.0 / (p.shape[0] + p.shape[1])) ** 0.5
                    p.uniform_(-bound, bound, generator=g)
                else:
                    p.zero_()
            self.flat_s.zero_()
            if self.table_state is not None:
                for t in self.table_state:
                    t.zero_()
        self.stream.synchronize()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        self.stream.synchronize()
        return {k: v.detach().cpu().clone() for k, v in self.params.items()}

    def _ensure_table_state(self):
        if self.table_state is None:
            with torch.cuda.stream(self.stream):
                self.table_state = [torch.zeros_like(t) for t in self.tables]

    # -------------------------------------------------------------------------------------------------------
    @_on_device
    def compile(self, choice, B: int, train: bool, clip: Optional[float] = 5.0, eps: float = 1e-2, graph: bool = False,
                grad_scale: Optional[float] = None, defer_dw: bool = True, row_grad_out: Optional[torch.Tensor] = None,
                local_optimizer: bool = True) -> CompiledPlan:
        """row_grad_out: storage [B * Fs * 16] the backward writes the per-sample embedding-row gradients into (a data-parallel step
        hands in the head of its all-gather send buffer: no copy between the backward and the exchange); local_optimizer = False: the
        plan gets no clip + Adagrad program of its own (a data-parallel step runs its optimizer over the GLOBAL batch)"""
        rgo = row_grad_out.data_ptr() if row_grad_out is not None else None
        fast = (id(choice), B, train, clip, eps, graph, grad_scale, defer_dw, rgo, local_optimizer)
        hit = self._last_plan
        if self.cfg.fixed and hit is not None and hit[0] == fast and hit[1] is choice:  # fixed sub-network, same choice object: skip the JSON key
            return hit[2]
        key = json.dumps([choice, B, train, clip, eps, graph, grad_scale, defer_dw, rgo, local_optimizer], sort_keys=True, default=_jsonable)
        if key in self._plans:
            self._last_plan = (fast, choice, self._plans[key])
            return self._plans[key]
        arena = None
        persist = bool(self.persist and self.level_schedule and self.cfg.fixed and B <= 256 and train and local_optimizer)
        if self.cfg.fixed and B <= 256 and train and self.level_schedule and (_UC_ARENA or persist):
            arena = Arena(self.device, uncached=True)  # (see _UC_ARENA; the persistent step needs it whatever the knob says)
        if not self.cfg.fixed:
            # Sampled paths rarely repeat: a small cache of plan slots, each with its own arena.  Evicting a plan needs no device
            # synchronisation: its slot is reused by a plan whose kernels are enqueued, in stream order, behind the evicted one's
            # (a caller that kept an autograd graph on the evicted plan is told so in backward: cp.evicted).
            if len(self._plans) >= 4:
                old = self._plans.pop(next(iter(self._plans)))
                old.evicted = True
                arena = old.arena
                old.arena = None
                self._last_plan = None
                # a plan is a web of closures over its context: cut it open so reference counting frees the ~10^4 objects now
                # instead of leaving them to the cyclic collector (whose full passes stalled a step for 17-56 ms)
                octx = getattr(old, "ctx", None)
                if octx is not None:
                    octx.closures = []
                    octx.deferred = []
                    octx.mha_reduce = []
                    octx.keep = []
                    octx.sk_workspace = None
            if arena is None:
                arena = self._spare_arenas.pop() if self._spare_arenas else Arena(self.device)
            arena.reset()
        cfg = self.cfg
        with torch.cuda.stream(self.stream):
            cp = CompiledPlan()
            cp.arena, cp.evicted = arena, False
            ctx = P.Ctx(B, self.device, self.params, self.grads, shape_only=False, train=train)
            ctx.sk_workspace = self._sk_workspace
            # parked weight-gradient batches are for one-launch-per-operator plans; the level scheduler places the products itself
            ctx.defer_dw = defer_dw and getattr(self, "park_weight_grads", True) and not (self.level_schedule and cfg.fixed and B <= 256)
            ctx.arena = arena
            ctx._pcache = self._pcache  # parameter name -> (shape, address): the arenas never move
            if (self.level_schedule and cfg.fixed and B <= 256) or getattr(self, "mha_bwd_form", 0) == 4:
                ctx.mha_bwd_form = 4
            cp.ctx = ctx
            new = (lambda n, dt=torch.float32: arena.alloc(n, dt).tensor()) if arena is not None else \
                (lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.device))
            cp.int_x = new(B * self.Fd).view(B, self.Fd)
            cp.cat_x = new(B * self.Fs, torch.int64).view(B, self.Fs)
            cp.y = new(B)
            cp.logits = new(B)
            cp.loss = new(1)
            cp.dlogits = new(B)
            int_buf = P.Buf(ctx, B * self.Fd, need_grad=False, tensor=cp.int_x)
            dense0 = P.DV(int_buf, 0, self.Fd, self.Fd)
            sbuf = ctx.buf(B * self.Fs * E)
            if row_grad_out is not None:
                assert row_grad_out.numel() == B * self.Fs * E and row_grad_out.dtype == torch.float32 and row_grad_out.is_contiguous()
                sbuf.g = row_grad_out
            sparse0 = P.SV(sbuf, 0, self.Fs, self.Fs * E)
            cp.sparse0 = sbuf
            ctx.raw_sparse = sbuf
            g = L.EmbedDesc()
            g.kind = L.OP_EMBED_GATHER
            g.B, g.Fs = B, self.Fs
            g.idx = cp.cat_x.data_ptr()
            if not self.host_embedding:
                for f in range(self.Fs):
                    g.table[f] = self.tables[f].data_ptr()
                    g.rows[f] = self.num_embeddings[f]
                g.out = sbuf.t.data_ptr()
                g.oob = self.oob.data_ptr()
            # the gather rides on the per-step staging launch (which holds the caller's id tensor anyway); cp.gather is the
            # stand-alone descriptor for the paths that stage by other means
            cp.gather = g
            st = L.StageDesc()
            st.kind = L.OP_STAGE_INPUTS
            st.B, st.Fd, st.Fs = B, self.Fd, self.Fs
            st.int_dst, st.cat_dst, st.y_dst = cp.int_x.data_ptr(), cp.cat_x.data_ptr(), cp.y.data_ptr()
            if not self.host_embedding:
                st.gather = g
            cp.stage = st
            d_last, s_last = P.network_walk(ctx, cfg, choice, dense0, sparse0)
            # final logit (supernet.py:592-598 / 657-664)
            fsegs, K = P.final_segments(d_last, s_last)
            w = ctx.param("_final.weight", (1, K))
            bptr = ctx.param("_final.bias", (1,))
            fd = L.FinalDesc()
            fd.kind = L.OP_FINAL_FWD
            fd.B, fd.nseg = B, len(fsegs)
            fd.w, fd.bias, fd.logits = w, bptr, cp.logits.data_ptr()
            for q, (s, ts) in enumerate(fsegs):
                fd.seg[q], fd.width[q], fd.ld[q], fd.off[q], fd.tok_stride[q] = s.view.ptr, s.width, s.view.ld, s.koff, ts
            ctx.emit(fd)
            if cfg.use_final_sigmoid:
                raise NotImplementedError("use_final_sigmoid is never enabled by the reference CLIs")
            scheduled = self.level_schedule and cfg.fixed and B <= 256
            cp.fwd_levels = cp.bwd_levels = None
            cp.dead_forward = []
            if train:  # (the backward program decides which forward results are read: built further down, before the packing)
                self._ensure_table_state()
                self._build_training_tail(cp, ctx, w, bptr, fsegs, B, K, grad_scale)
            fwd_list = list(ctx.fwd)
            if self.dead_code_elimination and cfg.fixed:
                fwd_list, cp.dead_forward = S.eliminate_dead_forward(fwd_list, ctx.bwd if train else [])
            if scheduled:
                fwd_descs, cp.fwd_levels = S.pack(fwd_list)
                self._resident_worklists(cp, fwd_descs)
            else:
                fwd_descs = fwd_list
            cp.fwd = Program(fwd_descs)
            cp.used_params = list(ctx.used_params)
            if train:
                bd, pre = cp.bce, cp._pre
                # The training programs fold BCEWithLogits into the final-logit backward (the first closure to run); bwd_core
                # keeps the plain one (d loss / d logits supplied by torch.autograd).
                fi = next(i for i, dsc in enumerate(ctx.bwd) if isinstance(dsc, L.FinalDesc))
                plain = ctx.bwd[fi]
                assert plain.kind == L.OP_FINAL_BWD
                fused = L.FinalDesc.from_buffer_copy(plain)
                fused.logits, fused.y, fused.loss, fused.dlogits_out = cp.logits.data_ptr(), cp.y.data_ptr(), cp.loss.data_ptr(), cp.dlogits.data_ptr()
                fused.grad_scale = bd.grad_scale
                cp.bce = bd
                bwd_descs = pre[1:] + ctx.bwd[:fi] + [fused] + ctx.bwd[fi + 1:]
                # forward + backward of a training step as ONE scheduled program: what the forward leaves off its critical path (a
                # block output no later block reads, second passes) runs beside the first backward levels
                # clip + Adagrad (built before the packing: the id-only half of its row dedup is an operator of the step like any other)
                odescs = []
                cp.dedup_ids, cp.ids_on_stage = None, False
                if local_optimizer:
                    odescs = self._optimizer_descs(cp, B, cp.cat_x, sbuf.grad_tensor() if (sbuf.grad_written and not self.host_embedding)
                                                   else None, clip, eps)
                ids_in_program = []
                if cp.dedup_ids is not None:
                    if scheduled and _IDS_AS_ITEM:
                        # ... which depends on nothing but the staged ids and is read by nothing before the optimizer: the level scheduler puts
                        # it where it costs least (beside a Transformer backward), off both the staging launch and the step's tail
                        ids_in_program = [cp.dedup_ids]
                    elif B <= 256:  # on the staging launch (which holds the caller's ids)
                        cp.stage.dedup_ids = cp.dedup_ids
                        cp.ids_on_stage = True
                    else:           # in-stream programs (large batch): in front of the launch that needs it
                        odescs = [cp.dedup_ids] + odescs
                cp.fb = None
                if scheduled:
                    # (alloc: a forward product with several levels of slack may be re-cut into split-K items — only in the JOINT
                    # program, where the backward's latency-bound levels are there to hide it)
                    jf, jb = self._fuse_final(fwd_list, bwd_descs, fused) if _FUSE_FINAL else (fwd_list, bwd_descs)
                    fb_descs = None
                    cp.persistent = False
                    cp.level_tuning = None
                    if persist:
                        try:
                            fb_descs, cp.fb_levels = S.pack_persistent(
                                ids_in_program + jf + jb, lambda nb: torch.empty(max(int(nb), 16), dtype=torch.uint8, device=self.device),
                                arena.ranges, alloc=ctx.alloc)
                            cp.persistent = True
                            self._persist_errs += [d.err_tensor for d in fb_descs if isinstance(d, L.PersistDesc)]
                        except S.PersistRefused as e:
                            import warnings
                            warnings.warn("persistent step refused (%s): one launch per level instead" % (e,))
                    if fb_descs is None:
                        fb_descs, cp.fb_levels = S.pack(ids_in_program + jf + jb, alloc=ctx.alloc)
                        if _WL_TUNE and S.BALANCE:
                            fb_descs, cp.level_tuning = self._tune_levels(ids_in_program + jf + jb, ctx.alloc, fb_descs)
                    self._resident_worklists(cp, fb_descs)
                    cp.fb = Program(fb_descs)
                    bwd_descs, cp.bwd_levels = S.pack(bwd_descs)
                    self._resident_worklists(cp, bwd_descs)
                cp.bwd = Program(bwd_descs)
                cp.bwd_tail_start = len(pre) - 1 + ctx.bwd_tail_start  # cp.bwd.descs[this:] = the parked weight-gradient products
                cp.bwd_marks = [(b, len(pre) - 1 + idx) for b, idx in ctx.block_marks]  # block b's gradients complete after cp.bwd.descs[:idx]
                cp.bwd_core = Program(pre[1:] + ctx.bwd)  # dlogits supplied by the caller (autograd path)
                # fine-tune-last-layer mode (SuperNet.set_mode_to_finelune_last_only, the searcher's candidate evaluation):
                # only d loss / d _final is wanted, i.e. the weight part of the final-logit backward and nothing else
                last = L.FinalDesc.from_buffer_copy(plain)
                for q in range(L.MAX_SEGS):
                    last.dseg[q] = None
                tail = [ctx.bwd[fi + 1]] if plain.nsplit > 1 else []  # (the partial-sum launch of a split final backward)
                cp.bwd_final_only = Program(pre[1:] + [last] + tail)
                cp.opt = Program(odescs)
                if graph:
                    cp.step = Program((cp.fb.descs if cp.fb is not None else cp.fwd.descs + cp.bwd.descs) + cp.opt.descs)
                    cp.step.capture(self.stream.cuda_stream)
            elif graph:
                cp.fwd.capture(self.stream.cuda_stream)
        if graph or arena is None or cfg.fixed:
            self.stream.synchronize()  # these plans are built (zero-filled, captured) on the private stream, and replayed on the caller's
        self._plans[key] = cp
        self._last_plan = (fast, choice, cp)
        return cp

    def _resident_worklists(self, cp, descs):
        """nasrec_worklist_prepare for every worklist launch of `descs` (a fixed plan: the pointers inside the items never change): the
        descriptor objects stay what they are (reports, tests, the data-parallel readiness model read them), a Program launches their
        `launch_as` — the device-resident form — instead"""
        if not _WL_RESIDENT:
            return
        lists = [d for d in descs if isinstance(d, L.WorklistDesc) and not hasattr(d, "launch_as")]
        if not lists:
            return
        lib = L.load()
        sz = (C.sizeof(L.WorklistDesc) + 255) & ~255
        buf = torch.empty(sz * len(lists), dtype=torch.uint8, device=self.device)
        if not hasattr(cp, "_wl_resident"):
            cp._wl_resident = []
        cp._wl_resident.append(buf)  # (owned by the plan)
        self.stream.synchronize()  # (the copies below are synchronous on the null stream)
        for i, d in enumerate(lists):
            dv = L.WorklistDevDesc()
            L.check(lib.nasrec_worklist_prepare(C.addressof(d), buf.data_ptr() + i * sz, C.addressof(dv)))
            d.launch_as = dv

    def _tune_levels(self, prog_in, alloc, default_descs):
        """The balancing pass of schedule.pack places slack operators by a duration MODEL whose least certain entries are the Transformer
        bodies (one workgroup per sample: how much of their time other items can share depends on what those items are).  A fixed
        sub-network's batch-256 plan is compiled once and run for an epoch, so the model's Transformer durations are treated as knobs:
        the joint program is packed for a handful of settings (schedule.TUNE_MHA_NS), every distinct schedule is replayed on the plan's
        own (zero-filled) buffers, and the fastest one wins if it beats the default by more than the timing noise.  Any schedule gives
        the same bits (same bodies, same operands, dependencies respected: tests/test_parity_gpu.py), so this only moves time.
        Returns (descriptors, report)."""
        sp = self.stream.cuda_stream

        def sig(descs):
            return hash(b"".join(bytes(d) for d in descs))

        def run_ms(prog, reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            for _ in range(reps):
                prog.run(sp)
            e1.record(self.stream)
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        keep = list(S._MHA_NS)
        cands = [("default", default_descs)]
        seen = {sig(default_descs)}
        try:
            for ns in S.TUNE_MHA_NS:
                S._MHA_NS[:] = list(ns) + keep[len(ns):]
                descs, _ = S.pack(prog_in, alloc=alloc)
                h = sig(descs)
                if h not in seen:
                    seen.add(h)
                    cands.append((",".join(str(v) for v in ns), descs))
        finally:
            S._MHA_NS[:] = keep
        if len(cands) == 1:
            return default_descs, {"candidates": 1, "chosen": "default"}
        progs = [(name, descs, Program(descs)) for name, descs in cands]
        for _, _, pg in progs:
            run_ms(pg, 5)  # descriptors and code warm
        times = {name: [] for name, _, _ in progs}
        for _ in range(5):  # interleaved rounds: clock and cache drift hits every candidate alike
            for name, _, pg in progs:
                times[name].append(run_ms(pg, 20))
        med = {name: sorted(v)[len(v) // 2] for name, v in times.items()}
        best = min(med, key=med.get)
        if med[best] > med["default"] * (1.0 - _WL_TUNE_MARGIN):
            best = "default"
        chosen = next(descs for name, descs, _ in progs if name == best)
        return chosen, {"candidates": len(progs), "chosen": best, "ms": {k: round(v, 5) for k, v in med.items()}}

    def _build_training_tail(self, cp, ctx, w, bptr, fsegs, B, K, grad_scale):
        """BCE descriptor, the final-logit backward closure, the backward program (ctx.bwd) and, for a supernet, the path's chunk
        table: everything the forward packing has to know about (which forward results the backward reads)"""
        cfg, arena = self.cfg, cp.arena
        bd = L.BceDesc()
        bd.kind = L.OP_BCE
        bd.B = B
        bd.grad_scale = grad_scale if grad_scale is not None else 1.0 / B
        bd.logits, bd.y, bd.loss, bd.dlogits = cp.logits.data_ptr(), cp.y.data_ptr(), cp.loss.data_ptr(), cp.dlogits.data_ptr()
        pre = [bd]

        def final_bwd():
            e = L.FinalDesc()
            e.kind = L.OP_FINAL_BWD
            e.B, e.nseg = B, len(fsegs)
            e.w, e.bias, e.dlogits = w, bptr, cp.dlogits.data_ptr()
            e.dw, e.dbias = self.grads["_final.weight"].data_ptr(), self.grads["_final.bias"].data_ptr()
            for q, (s, ts) in enumerate(fsegs):
                gp, acc = ctx.gtarget(s.view)
                e.seg[q], e.dseg[q], e.width[q], e.ld[q], e.off[q], e.dseg_accumulate[q] = s.view.ptr, gp, s.width, s.view.ld, s.koff, acc
                e.tok_stride[q] = ts
            if B > 256:
                # d loss / d _final over a large batch: nsplit batch slices in parallel -> partial [nsplit, K + 1], summed in
                # fixed order by the launch behind it (one workgroup per 16 columns walked 4096 rows in 132 us)
                e.nsplit = max(2, min(32, B // 128))
                part = ctx.alloc(e.nsplit * (K + 1))
                e.dw = part.data_ptr()
                r = L.ReduceRowsDesc()
                r.kind = L.OP_REDUCE_ROWS
                r.R, r.C, r.ld, r.in_ = e.nsplit, K + 1, K + 1, part.data_ptr()
                r.ndst = 2
                r.dst[0], r.dst_off[0], r.dst_len[0] = self.grads["_final.weight"].data_ptr(), 0, K
                r.dst[1], r.dst_off[1], r.dst_len[1] = self.grads["_final.bias"].data_ptr(), K, 1
                ctx.emit(e)
                ctx.emit(r)
                return
            ctx.emit(e)

        ctx.on_backward(final_bwd)
        ctx.build_backward()
        cp.chunk_tab, cp.nchunks, cp.path_spans = None, 0, None
        if not cfg.fixed:
            # Paths differ from step to step.  torch skips parameters whose grad is None (everything outside the path),
            # so zero_grad, the norm and Adagrad touch ONLY the arena ranges this path trains — the same arithmetic on
            # the path's share of the arena (a quarter of it for a `default` xlarge path).  Gradients left over from
            # other paths outside these ranges are never read.
            names = [n for n in list(ctx.grad_params) + ["_final.weight", "_final.bias"] if not n.startswith("_embedding.")]
            cp.path_spans = [(self.offsets[n], self.params[n].numel()) for n in dict.fromkeys(names)]
            flat = P.path_chunks(cp.path_spans)
            cp.nchunks = len(flat) // 2
            cp.chunk_tab = (arena.alloc(len(flat), torch.int64).tensor() if arena is not None
                            else torch.empty(len(flat), dtype=torch.int64, device=self.device))
            pre += P.const_i64_descs(cp.chunk_tab.data_ptr(), flat)
            ms = P.memset_desc(self.flat_g)
            ms.chunks, ms.nchunks = cp.chunk_tab.data_ptr(), cp.nchunks
            pre.append(ms)
        elif _FIXED_OPT_TABLE:
            # A fixed sub-network can hold parameters no gradient ever reaches (the reference leaves their grad None: in
            # ea_criteo_kaggle_xlarge_best_1shot.json all of block 5, 1.2 M of the 2.2 M dense parameters — no later block selects it).
            # Their gradient stays zero, and Adagrad with g = 0 changes neither state nor parameter: the norm and the update walk a chunk
            # table of the trained ranges only (half the bytes of the optimizer's bandwidth-bound pass).  Written once, here.
            names = [n for n in list(ctx.grad_params) + ["_final.weight", "_final.bias"] if not n.startswith("_embedding.")]
            spans = [(self.offsets[n], self.params[n].numel()) for n in dict.fromkeys(names)]
            if sum(n for _, n in spans) < 0.9 * self.flat_numel:
                flat = P.path_chunks(spans, chunk=1024)  # (small pieces: a workgroup per piece keeps ~10^3 workgroups on the pass)
                cp.nchunks = len(flat) // 2
                cp.chunk_tab = torch.empty(len(flat), dtype=torch.int64, device=self.device)
                lib = L.load()
                for dsc in P.const_i64_descs(cp.chunk_tab.data_ptr(), flat):
                    L.check(lib.nasrec_launch(self.stream.cuda_stream, C.addressof(dsc)))
                self.stream.synchronize()  # (the descriptors carry the values: they must outlive their launches)
        cp.bce, cp._pre = bd, pre

    def _optimizer_descs(self, cp, Bg, cat_x, sparse_grad, clip, eps, rank_layout=None):
        """clip_grad_norm_ + Adagrad (train_utils.py:285-286): row-sparse on the tables, flat on the dense arena.
        rank_layout (data-parallel step): (samples per rank, floats between the ranks' chunks) of `sparse_grad` when it is the receive
        buffer of an all-gather that carries more than the rows (parallel.py); None = one contiguous [Bg, Fs, 16] array.
        Leaves `cp.dedup_ids` (the id-only half of the row dedup, to be launched once the ids are in `cat_x`) or None."""
        descs = []
        cp.dedup_ids = None
        nb = (Bg + 255) // 256
        new = getattr(cp, "arena", None)
        new = (lambda n, dt=torch.float32: cp.arena.alloc(n, dt).tensor()) if new is not None else \
            (lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.device))
        nblk = max(1, min(256, (self.flat_numel + 256 * 8 - 1) // (256 * 8)))
        tab, ntab = getattr(cp, "chunk_tab", None), getattr(cp, "nchunks", 0)
        if tab is not None:
            nblk = max(1, min(256, ntab))
        if getattr(cp, "leader", None) is None:  # (the data-parallel optimizer shares these across the plans of a run)
            cp.leader = new(Bg * self.Fs, torch.int32)
            cp.gsum = new(Bg * self.Fs * E)
            cp.emb_partial = new(self.Fs * nb)
            cp.dense_partial = new(256)
        if sparse_grad is not None:
            dd = L.EmbDedupDesc()
            dd.kind = L.OP_EMB_DEDUP
            dd.B, dd.Fs = Bg, self.Fs
            dd.idx, dd.dout = cat_x.data_ptr(), sparse_grad.data_ptr()
            dd.leader, dd.gsum, dd.sumsq_partial = cp.leader.data_ptr(), cp.gsum.data_ptr(), cp.emb_partial.data_ptr()
            dd.overflow = self.oob.data_ptr() + 4
            descs.append(dd)
        sq = L.SumsqDesc()
        sq.kind = L.OP_SUMSQ
        sq.nblocks, sq.n = nblk, self.flat_numel
        sq.x, sq.partial = self.flat_g.data_ptr(), cp.dense_partial.data_ptr()
        if tab is not None:
            sq.chunks, sq.nchunks = tab.data_ptr(), ntab
        descs.append(sq)
        cc = L.ClipCoefDesc()
        cc.kind = L.OP_CLIP_COEF
        cc.n_a, cc.n_b = nblk, (self.Fs * nb if sparse_grad is not None else 0)
        cc.max_norm = float(clip) if clip is not None else 0.0
        cc.partial_a, cc.partial_b, cc.out = cp.dense_partial.data_ptr(), cp.emb_partial.data_ptr(), self.clip_out.data_ptr()
        descs.append(cc)
        ad = L.AdagradDenseDesc()
        ad.kind = L.OP_ADAGRAD_DENSE
        ad.eps, ad.n = eps, self.flat_numel
        ad.p, ad.g, ad.state = self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_s.data_ptr()
        ad.lr, ad.coef = self.lr_dev.data_ptr(), self.clip_out.data_ptr()
        if tab is not None:
            ad.chunks, ad.nchunks = tab.data_ptr(), ntab
        descs.append(ad)
        if sparse_grad is not None:
            ar = L.AdagradRowsDesc()
            ar.kind = L.OP_ADAGRAD_ROWS
            ar.B, ar.Fs, ar.eps = Bg, self.Fs, eps
            ar.idx, ar.leader, ar.gsum = cat_x.data_ptr(), cp.leader.data_ptr(), cp.gsum.data_ptr()
            for f in range(self.Fs):
                ar.table[f] = self.tables[f].data_ptr()
                ar.state[f] = self.table_state[f].data_ptr()
                ar.rows[f] = self.num_embeddings[f]
            ar.lr, ar.coef = self.lr_dev.data_ptr(), self.clip_out.data_ptr()
            descs.append(ar)
        if sparse_grad is not None:
            app = L.OptApplyDesc()
            app.kind = L.OP_OPT_APPLY
            # (2.2 M parameters of the batch-256 bench network: 256 / 512 / 1024 / 2048 / 4096 workgroups -> 15.3 / 10.9 / 8.6 / 9.5 / 10.8 us)
            cap = 1024 if self.flat_numel <= (4 << 20) else 2048
            app.dense_blocks = min(cap, ntab if tab is not None else (self.flat_numel + 255) // 256)
            app.clip, app.dense, app.rows = cc, ad, ar
            if Bg <= DEDUP_SPLIT_MAX_B:
                # Two halves (csrc/dedup_bodies.h): leaders, duplicate lists and their order depend on the ids only — cp.dedup_ids runs
                # as soon as the ids are there (B <= 256: on the staging launch; a data-parallel step: behind the ids all-gather, beside the
                # forward) — and one launch behind the backward sums the duplicate rows IN PLACE and squares what the clip needs.
                capn = 256
                while capn < Bg:
                    capn *= 2
                if getattr(cp, "dd_order", None) is None:
                    cp.dd_order = new(self.Fs * capn, torch.int32)
                    cp.dd_lists = new(self.Fs * capn, torch.int32)
                    cp.dd_counts = new(self.Fs * 2, torch.int32)
                    cp.dd_heads = new(self.Fs * capn, torch.int32) if capn > 256 else None
                row_blocks = max(1, min(512, (Bg * self.Fs * 4 + 255) // 256))
                if getattr(cp, "emb_partial2", None) is None or cp.emb_partial2.numel() < self.Fs + row_blocks:
                    cp.emb_partial2 = new(self.Fs + row_blocks)
                ids = L.DedupIdsDesc()
                ids.kind, ids.B, ids.Fs, ids.cap = L.OP_DEDUP_IDS, Bg, self.Fs, capn
                ids.idx, ids.leader = cat_x.data_ptr(), cp.leader.data_ptr()
                ids.order, ids.lists, ids.counts = cp.dd_order.data_ptr(), cp.dd_lists.data_ptr(), cp.dd_counts.data_ptr()
                ids.heads = cp.dd_heads.data_ptr() if cp.dd_heads is not None else None
                cp.dedup_ids = ids
                r2 = L.OptReduce2Desc()
                r2.kind, r2.B, r2.Fs, r2.cap = L.OP_OPT_REDUCE2, Bg, self.Fs, capn
                r2.rank_B, r2.rank_stride = rank_layout if rank_layout else (0, 0)
                r2.row_blocks = row_blocks
                r2.rows, r2.leader = sparse_grad.data_ptr(), cp.leader.data_ptr()
                r2.order, r2.lists, r2.counts, r2.heads = ids.order, ids.lists, ids.counts, ids.heads
                r2.sumsq_partial = cp.emb_partial2.data_ptr()
                r2.sumsq = sq
                app.clip.partial_b, app.clip.n_b = cp.emb_partial2.data_ptr(), self.Fs + row_blocks
                app.rows.gsum = sparse_grad.data_ptr()  # summed in place: a leader's row holds its sum
                app.rows.rank_B, app.rows.rank_stride = r2.rank_B, r2.rank_stride
                return [r2, app]
            if rank_layout:  # (the one-launch kernels read the rows where the all-gather left them; their sums go to the contiguous gsum)
                dd.rank_B, dd.rank_stride = rank_layout
            if Bg <= 256:
                # the five launches collapse into the two that the grid-wide dependencies require
                red = L.OptReduceDesc()
                red.kind = L.OP_OPT_REDUCE
                red.dedup, red.sumsq = dd, sq
                return [red, app]
            return [dd, sq, app]  # global batch of a data-parallel step: chunked dedup + merge (two launches), then the same apply
        return descs

    @_on_device
    def reserve(self, B: int, train: bool = True, choice=None, freeze_gc: bool = False):
        """Size every plan slot for the LARGEST path at batch B (default: the warm-up choice = the full path) and allocate the
        engine-level workspaces, so that no sampled path of a run has to grow an arena (an allocator call + device
        synchronisation in the middle of a step: 1-2 ms spikes in the per-step times).  Host work only; nothing is launched.
        freeze_gc (opt-in, for a process that keeps ONE engine for its whole life — bench.py, the train_supernet CLI): move
        everything alive now into the cyclic collector's permanent generation, so that its full passes walk the few plans in flight
        instead of ~10^6 long-lived objects (17-56 ms stalls per step otherwise).  Frozen objects are never collected: a process that
        builds engine after engine (eval_subnet_from_supernet: one SuperNet per candidate) must leave it off."""
        if self.cfg.fixed:
            return 0
        cp = self.compile(choice if choice is not None else self.warm_choice, B, train)
        need = cp.arena.used_bytes() + Arena.CHUNK
        key = next(k for k, v in self._plans.items() if v is cp)
        self._plans.pop(key)
        cp.evicted = True
        self._last_plan = None
        arenas = [cp.arena] + list(self._spare_arenas)
        cp.arena = None
        # the warm plan is a web of closures over its context, which holds the parameter / gradient dicts, the arena and a bound
        # method of this engine: cut it open so that reference counting frees it now (and nothing of it is frozen below)
        wctx = getattr(cp, "ctx", None)
        if wctx is not None:
            wctx.closures, wctx.deferred, wctx.mha_reduce, wctx.keep = [], [], [], []
            wctx.sk_workspace = None
            wctx.arena = None
            wctx.params = wctx.grads = None
        cp.ctx = None
        cp = wctx = None  # (not `del`: `cp` is read by the generator expression above, and the module also builds under Cython)
        live = len(self._plans)
        while len(arenas) + live < 4:
            arenas.append(Arena(self.device))
        for a in arenas:
            a.reserve(need)
        for v in self._plans.values():
            if v.arena is not None:
                v.arena.reserve(need)
        self._spare_arenas = arenas
        self._sk_workspace()
        if freeze_gc:
            import gc
            gc.collect()
            gc.freeze()
        return need

    def _sk_workspace(self):
        """partial-tile workspace of the balanced GEMM schedule (100 MB): one per engine — the launches of a program run in
        stream order, and each one's second pass has consumed the buffer before the next launch writes it"""
        if getattr(self, "_sk_ws", None) is None:
            self._sk_ws = torch.empty(L.SK_WORKSPACE_FLOATS, dtype=torch.float32, device=self.device)
        return self._sk_ws

    # -------------------------------------------------------------------------------------------------------
    def _sp(self):
        """Launches go to the CALLER's current stream (no cross-stream event pair per step: each one costs a barrier packet
        and ~10 us of GPU idle); the private stream only builds and captures plans."""
        return torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _fuse_final(fwd_list, bwd_descs, fused):
        """joint forward + backward program: the final logit's forward and the per-sample part of its backward (d loss / d features =
        (sigmoid(logit) - y) / B * w: supernet.py:592-598 under BCEWithLogitsLoss) become ONE operator (NASREC_OP_FINAL_FUSED: the
        wavefront that sums a sample's logit writes its feature gradients too), so the backward pass starts one launch boundary
        earlier; the parts that need every sample's logit (d w, d bias, the loss) stay a FINAL_BWD with dseg_done, which nothing
        waits for.  Same expressions per element: the bit-identity tests against one launch per operator cover it."""
        fi = [i for i, d in enumerate(fwd_list) if isinstance(d, L.FinalDesc) and d.kind == L.OP_FINAL_FWD]
        if len(fi) != 1 or fused.nsplit > 1 or not any(d is fused for d in bwd_descs):
            return fwd_list, bwd_descs
        f = fwd_list[fi[0]]
        same = f.B == fused.B and f.nseg == fused.nseg and f.w == fused.w and f.logits == fused.logits and all(
            f.seg[q] == fused.seg[q] and f.width[q] == fused.width[q] and f.ld[q] == fused.ld[q] and f.off[q] == fused.off[q]
            and f.tok_stride[q] == fused.tok_stride[q] and (not fused.dseg[q] or f.seg[q]) for q in range(f.nseg))
        if not same:
            return fwd_list, bwd_descs
        both = L.FinalDesc.from_buffer_copy(fused)
        both.kind, both.bias = L.OP_FINAL_FUSED, f.bias
        rest = L.FinalDesc.from_buffer_copy(fused)
        rest.dseg_done = 1
        return ([both if i == fi[0] else d for i, d in enumerate(fwd_list)], [rest if d is fused else d for d in bwd_descs])

    def _stage_inputs(self, sp, cp, int_x, cat_x, y=None, lr=None, rows=None, launch=True):
        """one launch: batch -> the plan's static buffers (+ this step's learning rate -> device scalar); host_embedding: the
        looked-up rows [B, Fs, 16] come with the batch.  launch=False: only patch the prebuilt staging descriptor (the caller launches it
        as the head of a program) — returns True when that is what happened, False when the inputs went another way and are staged"""
        if self.host_embedding:
            if rows is None:
                raise L.EngineError("this engine holds no embedding table (place_embedding_on_cpu): pass the looked-up rows")
            cp.sparse0.t.copy_(rows.reshape(-1).to(torch.float32), non_blocking=True)
        ok = (int_x.is_cuda and cat_x.is_cuda and int_x.dtype == torch.float32 and cat_x.dtype == torch.int64
              and int_x.is_contiguous() and cat_x.is_contiguous()
              and (y is None or (y.is_cuda and y.dtype == torch.float32 and y.is_contiguous())))
        if not ok:  # host tensors / other dtypes: let torch convert and copy
            cp.int_x.copy_(int_x.reshape(cp.int_x.shape), non_blocking=True)
            cp.cat_x.copy_(cat_x.reshape(cp.cat_x.shape), non_blocking=True)
            if y is not None:
                cp.y.copy_(y.reshape(cp.y.shape), non_blocking=True)
            if lr is not None:
                self.lr_dev.fill_(float(lr))
            if not self.host_embedding:
                L.check(L.load().nasrec_launch(sp, C.addressof(cp.gather)))
            if getattr(cp, "ids_on_stage", False):  # (the staging launch would have carried it)
                L.check(L.load().nasrec_launch(sp, C.addressof(cp.dedup_ids)))
            return False
        d = cp.stage  # prebuilt per plan: only the sources change from step to step
        d.int_src, d.cat_src = int_x.data_ptr(), cat_x.data_ptr()
        d.y_src = y.data_ptr() if y is not None else None
        if lr is not None:
            d.lr, d.lr_dst = float(lr), self.lr_dev.data_ptr()
        else:
            d.lr_dst = None
        if not launch:
            return True
        L.check(L.load().nasrec_launch(sp, C.addressof(d)))
        return False

    @_on_device
    def forward(self, int_x, cat_x, choice=None, graph=False, rows=None):
        """logits [B,1] for the given choice (fixed mode: the fixed choice)."""
        choice = choice if choice is not None else self.warm_choice
        B = int(int_x.shape[0])
        cp = self.compile(choice, B, train=False, graph=graph)
        sp = self._sp()
        self._stage_inputs(sp, cp, int_x, cat_x, rows=rows)
        if graph:
            cp.fwd.replay(sp)
        else:
            cp.fwd.run(sp)
        return cp.logits.view(B, 1)

    def prefers_graph(self, B: int) -> bool:
        """graph = None: replay the captured step, or launch its program?  A level-scheduled step (fixed sub-network, batch <= 256) is
        ~27 launches issued by ONE host call (nasrec_program_run: ~3 us each, the host stays far ahead of a 0.25 ms step), and every
        graph launch ends on a ~8.5 us bubble before the next command starts (rocprofv3 kernel trace, tools/launch_gaps.py: graph's last
        kernel -> next launch 8.4 - 8.7 us, 0.0 between eagerly launched kernels): 0.2575 ms replayed, 0.2515 ms launched.  Plans
        of 60+ launches (no level schedule; larger batches) keep the graph: there the host call is what bounds the step.  (A host that
        has more to do between steps — the training harness with its data pipe — is better off replaying: SuperNet.engine_train_step.)"""
        return not (self.level_schedule and self.cfg.fixed and B <= 256) or os.environ.get("NASREC_STEP_GRAPH", "auto") == "1"

    @_on_device
    def train_step(self, int_x, cat_x, y, lr: float, choice=None, clip: Optional[float] = 5.0, eps: float = 1e-2, graph: Optional[bool] = False,
                   staged: bool = False):
        """zero_grad -> forward -> BCE -> backward -> clip_grad_norm_ -> Adagrad (train_utils.py:262-286).
        Returns the (device) loss tensor of this step.  `staged`: inputs are already in the plan's static buffers.  graph: True = replay
        the captured step, False = launch its program, None = whichever is faster for this plan (prefers_graph)."""
        choice = choice if choice is not None else self.warm_choice
        if graph is None:
            graph = self.cfg.fixed and self.prefers_graph(int(int_x.shape[0]) if int_x is not None else int(self._last_plan[2].cat_x.shape[0]))
        if self.host_embedding:
            raise L.EngineError("the fused training step needs the embedding tables on the device (place_embedding_on_cpu keeps them "
                                "on the host): use the nn.Module path (forward / backward / torch optimizer)")
        if int_x is not None:
            B = int(int_x.shape[0])
        else:  # pre-staged inputs: the batch size is the one of the plan they were staged into
            assert staged and self._last_plan is not None, "train_step without inputs needs a previously compiled plan holding them"
            B = int(self._last_plan[2].cat_x.shape[0])
        cp = self.compile(choice, B, train=True, clip=clip, eps=eps, graph=graph)
        sp = self._sp()
        if not staged:
            whole = not graph and getattr(cp, "fb", None) is not None and not self.host_embedding
            if self._stage_inputs(sp, cp, int_x, cat_x, y, lr, launch=not whole):
                # a launched (not replayed) step: staging launch + joint program + optimizer as ONE host call
                if getattr(cp, "whole", None) is None:
                    cp.whole = Program([cp.stage] + list(cp.fb.descs) + list(cp.opt.descs))
                cp.whole.run(sp)
                return cp.loss
        else:
            self.lr_dev.fill_(float(lr))
            L.check(L.load().nasrec_launch(sp, C.addressof(cp.gather)))
            if getattr(cp, "ids_on_stage", False):
                L.check(L.load().nasrec_launch(sp, C.addressof(cp.dedup_ids)))
        if graph:
            cp.step.replay(sp)
        else:
            if getattr(cp, "fb", None) is not None:
                cp.fb.run(sp)
            else:
                cp.fwd.run(sp)
                cp.bwd.run(sp)
            cp.opt.run(sp)
        return cp.loss

    @_on_device
    def run_forward(self, cp, int_x, cat_x, rows=None):
        """forward program of an already compiled (training) plan; logits land in cp.logits"""
        sp = self._sp()
        self._stage_inputs(sp, cp, int_x, cat_x, rows=rows)
        cp.fwd.run(sp)

    @_on_device
    def run_backward(self, cp, dlogits, final_only: bool = False):
        """backward program with an externally supplied d(loss)/d(logits) [B,1] (torch.autograd entry)"""
        cp.dlogits.copy_(dlogits.reshape(-1), non_blocking=True)
        (cp.bwd_final_only if final_only else cp.bwd_core).run(self._sp())

    @_on_device
    def forward_backward(self, int_x, cat_x, y, choice=None, grad_scale=None):
        """forward + BCE + backward only (gradients left in self.grads / plan.sparse0 gradient); used by the data-parallel
        wrapper and by the parity tests."""
        choice = choice if choice is not None else self.warm_choice
        B = int(int_x.shape[0])
        cp = self.compile(choice, B, train=True, grad_scale=grad_scale)
        sp = self._sp()
        self._stage_inputs(sp, cp, int_x, cat_x, y)
        if getattr(cp, "fb", None) is not None:
            cp.fb.run(sp)
        else:
            cp.fwd.run(sp)
            cp.bwd.run(sp)
        return cp

    def check_indices(self):
        """raise IndexError if any embedding id seen so far was out of range (torch raises at lookup time)"""
        torch.cuda.synchronize(self.device)
        for t in self._persist_errs:
            e = t.view(torch.int32)[:3].tolist()
            if e[0] != 0:
                raise RuntimeError("persistent step: a workgroup of item %d waited more than 10 ms for item %d (results of that step are invalid)" % (e[1], e[2]))
        flags = self.oob.tolist()
        if flags[0] != 0:
            raise IndexError("index out of range in embedding lookup")
        if flags[1] != 0:
            raise RuntimeError("embedding-gradient merge: a hash partition overflowed (more than 16384 distinct ids hashed into one "
                               "partition of 4096 expected); the optimizer step that raised the flag must be discarded")


def _jsonable(o):
    if hasattr(o, "tolist"):
        return o.tolist()
    if hasattr(o, "item"):
        return o.item()
    raise TypeError(type(o))
