"""Plan compiler: (network description, choice, batch size) -> launch program for the HIP engine.

One walk over the choice (macro wiring supernet.py:513-668, block wiring :1067-1242, operators modules.py) is used
three ways:
  * shape inference  — which parameters exist and their shapes (LazyLinear semantics + the "delete the projection if
    the input already has the target width" rules, modules.py:344-364,389,491,586,743; supernet.py:1144,1225);
  * forward program  — a list of C-ABI descriptors (include/nasrec_hip.h) executed by nasrec_program_run;
  * backward program — emitted by replaying per-op closures in reverse, with liveness (dead branches are skipped, as
    autograd would) and write-vs-accumulate tracking per gradient buffer.

Concatenations are never materialised: every consumer takes a list of segments.  In supernet mode an un-chosen input
is a zero segment (supernet.py:536-568): it is dropped from the segment list while its width still advances the K
offset, so the arithmetic is identical to multiplying by zeros.
"""
import ctypes as C
import math
from typing import Dict, List, Optional

import torch

from . import _lib as L

NUM_MHA_HEADS = 8  # modules.py:26
DS_INTERACT_NUM_SPLITS = 8  # supernet.py:882
E = 16

DENSE_UNARY = ("linear-2d", "zeros-2d")
DENSE_BINARY = ("sum", "sigmoid-gating")
DENSE_SPARSE = ("dot-product",)
SPARSE_NODES = ("zeros-3d", "transformer", "linear-3d")

MHA_LEAVES = ["_mha.in_proj_weight", "_mha.in_proj_bias", "_mha.out_proj.weight", "_mha.out_proj.bias", "_attn_ln.weight",
              "_attn_ln.bias", "attn_fc1.weight", "attn_fc1.bias", "attn_fc2.weight", "attn_fc2.bias", "_attn_fc_ln.weight",
              "_attn_fc_ln.bias"]
MHA_SHAPES = [(48, 16), (48,), (16, 16), (16,), (16,), (16,), (16, 16), (16,), (16, 16), (16,), (16,), (16,)]
MHA_OFFS = [0, 768, 816, 1072, 1088, 1104, 1120, 1376, 1392, 1648, 1664, 1680]


class NetConfig:
    """The SuperNet constructor arguments that shape the computation (supernet.py:214-236)."""

    def __init__(self, num_blocks, ops_config, use_layernorm, activation="relu", embedding_dim=16, fixed=False,
                 last_n_blocks_out=1, use_final_sigmoid=False):
        assert embedding_dim == E, "the engine is specialised for embedding_dim == 16 (supernet.py:224)"
        if not 1 <= int(last_n_blocks_out) <= L.MAX_SEGS // 2:
            raise NotImplementedError("last_n_blocks_out = %s: the final-logit descriptor holds %d segments (dense + sparse of at most %d blocks)"
                                      % (last_n_blocks_out, L.MAX_SEGS, L.MAX_SEGS // 2))
        self.last_n_blocks_out = int(last_n_blocks_out)
        self.num_blocks = num_blocks
        self.ops_config = ops_config
        self.use_layernorm = bool(use_layernorm)
        self.activation = activation
        self.fixed = fixed
        self.use_final_sigmoid = use_final_sigmoid

    def block_ops(self, i):
        return self.ops_config[i] if isinstance(self.ops_config, list) else self.ops_config


# ----------------------------------------------------------------------------------------------------------------
# buffers and views
# ----------------------------------------------------------------------------------------------------------------
class Buf:
    def __init__(self, ctx, numel, need_grad=True, tensor=None):
        self.ctx = ctx
        self.numel = numel
        self.t = tensor if tensor is not None else ctx.alloc(numel)
        self.g = None
        self.need_grad = need_grad
        self.written = []  # column intervals [c0, c1) of each row whose gradient has been written

    @property
    def grad_written(self):
        return bool(self.written)

    def mark(self, c0=0, c1=1 << 30):
        self.written.append((c0, c1))

    def overlaps(self, c0, c1):
        return any(a < c1 and c0 < b for a, b in self.written)

    def grad_tensor(self):
        if self.g is None:
            self.g = self.ctx.alloc(self.numel)
        return self.g


class DV:
    """dense view [B, width], row stride ld, starting `off` floats into buf"""

    def __init__(self, buf, off, width, ld, padded=False):
        self.buf, self.off, self.width, self.ld = buf, off, width, ld
        self.padded = padded  # ld > width only to align the rows: the view still covers every column anybody reads

    @property
    def ptr(self):
        return self.buf.t.data_ptr() + 4 * self.off

    @property
    def gptr(self):
        return self.buf.grad_tensor().data_ptr() + 4 * self.off

    def full(self, B):
        if self.padded:
            return self.off == 0 and self.ld * B == self.buf.numel
        return self.off == 0 and self.width == self.ld and self.width * B == self.buf.numel

    def cols(self):
        c0 = self.off % self.ld
        return c0, c0 + self.width


class SV:
    """sparse view [B, N, 16] = tokens of a [B, Ntot, 16] slab; batch stride ld"""

    def __init__(self, buf, off, N, ld):
        self.buf, self.off, self.N, self.ld = buf, off, N, ld

    @property
    def ptr(self):
        return self.buf.t.data_ptr() + 4 * self.off

    @property
    def gptr(self):
        return self.buf.grad_tensor().data_ptr() + 4 * self.off

    def dense(self):
        return DV(self.buf, self.off, self.N * E, self.ld)

    def rows(self, n0, n):
        return SV(self.buf, self.off + n0 * E, n, self.ld)

    def full(self, B):
        return self.off == 0 and self.N * E == self.ld and self.N * E * B == self.buf.numel

    def cols(self):
        c0 = self.off % self.ld
        return c0, c0 + self.N * E


class Seg:
    """one segment of a virtual concatenation: view (or None = zeros) placed at offset koff of the consumer's K axis"""

    def __init__(self, view, koff, width):
        self.view, self.koff, self.width = view, koff, width


# ----------------------------------------------------------------------------------------------------------------
# compile context
# ----------------------------------------------------------------------------------------------------------------
class Ctx:
    def __init__(self, B, device, params: Optional[Dict[str, torch.Tensor]], grads: Optional[Dict[str, torch.Tensor]],
                 shape_only=False, train=True):
        self.B = B
        self.device = device
        self.params = params
        self.grads = grads
        self.shape_only = shape_only
        self.train = train
        self.shapes: Dict[str, tuple] = {}
        self.fwd: List = []
        self.bwd: List = []
        self.closures: List = []
        self.keep: List = []  # tensors kept alive for the lifetime of the plan
        self.used_params: List[str] = []
        self._used_set = set()
        self._pcache: Dict[str, tuple] = {}  # name -> (shape, data_ptr): one lookup per parameter use
        self.grad_params: List[str] = []  # parameters whose gradient the backward program writes
        self._grad_set = set()
        self.out = self.fwd  # current emission target
        self.deferred: List = []  # weight-gradient products parked until the end of the backward program
        self.bwd_tail_start = 0
        self.mha_reduce: List = []  # (MHA backward descriptor, [(grad ptr, column offset, length)]) awaiting the shared reduction
        self.ln_reduce: List = []  # LayerNorm parameter-gradient reductions of a large-batch plan awaiting their shared launch
        self.defer_dw = True
        self.arena = None  # optional Arena (engine.py): where alloc() takes its buffers from
        self.raw_sparse = None  # Buf of the embedding stem's output [B, Fs, 16] (set by the engine): see _flush_raw_dx
        self.deferred_raw: List = []
        self.sk_workspace = None  # optional callable -> tensor of L.SK_WORKSPACE_FLOATS floats shared by the plan's balanced GEMM launches
        self.mha_bwd_form = 0  # nasrec_mha_desc_t.bwd_form of the Transformer backward launches (4: the form worklist launches run)
        self.block_marks: List = []  # (block index, length of the backward program once that block's gradients are complete)

    # -- memory -----------------------------------------------------------------------------------------------
    def alloc(self, numel):
        if self.shape_only:
            return _FakeTensor()
        if self.arena is not None:  # bump allocation from the plan slot's arena: no allocator call per buffer
            return self.arena.alloc(int(numel))
        t = torch.empty(int(numel), dtype=torch.float32, device=self.device)
        self.keep.append(t)
        return t

    def zeros_i32(self, n):
        if self.shape_only:
            return _FakeTensor()
        t = torch.zeros(int(n), dtype=torch.int32, device=self.device)
        self.keep.append(t)
        return t

    def buf(self, numel, need_grad=True):
        return Buf(self, int(numel), need_grad)

    # -- parameters -------------------------------------------------------------------------------------------
    def param(self, name, shape, used=True):
        hit = self._pcache.get(name)
        if hit is not None and hit[0] == shape:  # (the common case of a re-compiled path: same name, same shape object contents)
            if used and name not in self._used_set:
                self._used_set.add(name)
                self.used_params.append(name)
            return hit[1]
        shape = tuple(int(s) for s in shape)
        if name in self.shapes:
            assert self.shapes[name] == shape, (name, self.shapes[name], shape)
        self.shapes[name] = shape
        if self.shape_only:
            return 0
        p = self.params[name]
        assert tuple(p.shape) == shape, "parameter %s has shape %s, the path needs %s" % (name, tuple(p.shape), shape)
        if used and name not in self._used_set:
            self._used_set.add(name)
            self.used_params.append(name)
        ptr = p.data_ptr()
        self._pcache[name] = (shape, ptr)
        return ptr

    def gparam(self, name):
        if name not in self._grad_set:
            self._grad_set.add(name)
            self.grad_params.append(name)
        return self.grads[name].data_ptr()

    # -- emission ---------------------------------------------------------------------------------------------
    def emit(self, desc):
        if not self.shape_only:
            self.out.append(desc)

    def on_backward(self, fn):
        if self.train and not self.shape_only:
            self.closures.append(fn)

    def build_backward(self):
        self.out = self.bwd
        for fn in reversed(self.closures):
            fn()
        _flush_raw_dx(self)
        _flush_mha_reduce(self)
        _flush_ln_reduce(self)
        self.bwd_tail_start = len(self.bwd)  # from here on: only the parked weight-gradient products
        _flush_deferred(self)
        self.out = self.fwd

    # gradient destination of a view: returns (ptr, accumulate) or (None, 0) if no gradient is wanted
    def gtarget(self, view):
        buf = view.buf
        if not buf.need_grad:
            return None, 0
        if (self.deferred or self.deferred_raw) and buf.g is not None and not self.shape_only:
            # A parked weight-gradient product reads this gradient storage as its dz operand, and a write is about to
            # land in it (gradient storage shared through _alias_add: the pre-FM block output takes further
            # contributions after the FM projection consumed the sum): issue those products first.
            lo = buf.g.data_ptr()
            hi = lo + 4 * buf.numel
            hit = [e for e in self.deferred if lo <= e[4]["A"] < hi]
            if hit:
                self.deferred = [e for e in self.deferred if not (lo <= e[4]["A"] < hi)]
                _flush_deferred(self, hit)
            if any(lo <= e["B"] < hi for e in self.deferred_raw):
                _flush_raw_dx(self)
        acc = buf.grad_written
        if not acc and not view.full(self.B):
            self.emit(memset_desc(buf.grad_tensor()))
            acc = True
        buf.mark(*view.cols())
        return view.gptr, int(acc)

    def live(self, view):
        """does any gradient reach this view? (column-interval granularity: e.g. the dense->sparse projection rows of
        a block output are dead when only the block's own DeepFM touched the slab's gradient)"""
        return view.buf.overlaps(*view.cols())


class _FakeTensor:
    def data_ptr(self):
        return 0


# ----------------------------------------------------------------------------------------------------------------
# descriptor builders
# ----------------------------------------------------------------------------------------------------------------
def memset_desc(t):
    d = L.MemsetDesc()
    d.kind = L.OP_MEMSET
    d.bytes = t.numel() * 4
    d.ptr = t.data_ptr()
    return d


def path_chunks(ranges, chunk=None):
    """[(offset, numel)] arena ranges (16-byte aligned offsets) -> flat [off0, n0, off1, n1, ...] chunk table of pieces
    <= NASREC_CHUNK_ELEMS elements: abutting ranges merge, nothing outside the ranges is covered"""
    chunk = chunk or L.CHUNK_ELEMS
    merged = []
    for off, n in sorted(ranges):
        if n <= 0:
            continue
        if merged and off <= merged[-1][0] + merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], off + n - merged[-1][0])
        else:
            merged.append([off, n])
    flat = []
    for off, n in merged:
        for o in range(0, n, chunk):
            flat += [off + o, min(chunk, n - o)]
    return flat


def const_i64_descs(dst_ptr, values):
    """NASREC_OP_CONST_I64 launches that write `values` to the int64 array at dst_ptr"""
    out = []
    for i in range(0, len(values), L.CONST_I64_MAX):
        part = values[i:i + L.CONST_I64_MAX]
        d = L.ConstI64Desc()
        d.kind = L.OP_CONST_I64
        d.n = len(part)
        d.dst = dst_ptr + 8 * i
        d.vals[:len(part)] = part
        out.append(d)
    return out


import os as _os
_SPLITK_CAP = int(_os.environ["NASREC_SPLITK_CAP"]) if _os.environ.get("NASREC_SPLITK_CAP") else None


def _splitk_for(tiles_mn, ktiles, nprob=1):
    """pick a split-K factor so that the launch stays within one wave of workgroups (256 CUs: `tools/splitk_sweep.py` shows
    a cliff right above it) with >= 64 k per split (one staged tile of the latency-regime configuration)"""
    if tiles_mn * nprob >= 192 or ktiles < 8:
        return 1
    if _SPLITK_CAP is not None:  # experiment knob (tools/ab_bench.sh): NASREC_SPLITK_CAP=1 disables split-K
        return max(1, min(_SPLITK_CAP, ktiles // 2, max(1, 256 // max(1, tiles_mn * nprob))))
    s = min(ktiles // 2, max(1, 256 // max(1, tiles_mn * nprob)))
    return max(1, min(s, 32))


GEMM_FAST_MIN_TILES = 120  # csrc/gemm_tile.h NASREC_GEMM_FAST_MIN_TILES
_FAST_MIN_K = int(_os.environ.get("NASREC_FAST_MIN_K", "1"))     # A/B knobs (csrc/gemm_fast.hip reads the first one too).  64 -> 1 in round 4: a [B,16] x [16,1024] product is a streaming
# write of B x 1024 floats, and the throughput kernel's epilogue (full 128-byte rows per store instruction) is the better store path whatever K is: cfg 5 +1.9 %, cfg 3 / 4 +1.1 % (A/B)
_LN_NBLK = int(_os.environ.get("NASREC_LN_NBLK", "512"))          # workgroups (of four row-per-wave wavefronts) of the dense LayerNorm backward (A/B knob; 256 until round 4:
# one workgroup per CU = four wavefronts per CU for a streaming kernel whose rows wait for the previous row's stores: cfg 5 +1.6 %, cfg 3 +0.7 %; 1024 = 512)
_TOKDW_CAP = int(_os.environ.get("NASREC_TOKDW_CAP", "128"))     # workgroups per problem of the token-axis weight gradient (A/B knob; 32 until round 4: a launch of two problems ran on 64 of
# the 256 CUs — 64x72x131072 x 2: 108.9 us at 22 TFLOP/s; cfg 5 +1.5 %, cfg 3 +0.6 %)
_FAST_MIN_KT = int(_os.environ.get("NASREC_FAST_MIN_KT", "4"))   # k-tiles of 32 per split of the throughput kernel (8 -> 4, round 4: a 4096x128x1024 product ran on 128 of 256 CUs; cfg 3 5.957 -> 5.930 ms, A/B)


def _fast_gemm_splitk(amode, bmode, cmode, segs, zmode):
    """Mirror of csrc/gemm_fast.hip `gemm_fast_eligible`: does this launch take the throughput-regime kernel (128x128x32
    tiles), and with which split-K?  -> S (>= 1) or None.  A product with few output tiles but a deep K (the weight gradients
    of a large batch: K = B) is split so that tiles x S fills the chip, >= 8 k-tiles of 32 per split."""
    if cmode != L.CM_PLAIN or (amode, bmode) not in _FAST_BINDINGS:
        return None
    kmax = 0
    kt_max = kt_sum = 0
    any_live = False
    for sd in segs:
        if sd.get("Aaux") or sd.get("Baux"):
            return None
        if not sd.get("A"):
            continue
        any_live = True
        K = sd["K"]
        if K > kmax:
            kmax = K
        # 31-bit byte offsets of the staging loads: operand extents stay below 2^29 floats (gemm_fast.hip:623-625)
        r, ld = max(sd["M"], sd["N"]), max(sd.get("lda", 0), sd.get("ldb", 0))
        if r * ld + K >= (1 << 29) or K * ld + r >= (1 << 29):
            return None
        kt = (K + 31) // 32
        kt_sum += kt
        if kt > kt_max:
            kt_max = kt
    if not any_live or kmax < _FAST_MIN_K:
        return None
    tiles = area = 0
    for sd in (segs if zmode else segs[:1]):
        tiles += ((sd["M"] + 127) // 128) * ((sd["N"] + 127) // 128)
        area += sd["M"] * sd["N"]
    if 2 * area < tiles * 128 * 128:
        return None  # skinny problems: the small-tile kernel
    kt = kt_max if zmode else kt_sum
    S = 1
    if tiles < 256:
        S = max(1, min(-(-256 // tiles), kt // _FAST_MIN_KT, 32))
    return S if tiles * S >= GEMM_FAST_MIN_TILES else None


_FAST_BINDINGS = ((L.AM_KC, L.AM_KC), (L.AM_KC, L.AM_RC), (L.AM_RC, L.AM_RC))
def kslice_eligible(amode, bmode, cmode, segs, zmode):
    """mirror of csrc/gemm_kslice.hip `gemm_kslice_eligible`: one large forward product of the batch-256 regime runs as a single pass
    with K split inside the workgroup (no split-K workspace, no second launch)"""
    if (amode, bmode, cmode) != (L.AM_KC, L.AM_KC, L.CM_PLAIN) or zmode:
        return False
    M, N = segs[0]["M"], segs[0]["N"]
    K = 0
    for sd in segs:
        if sd.get("Aaux") or sd.get("Baux") or sd.get("ones_col") or (0 < sd.get("Mvalid", M) < M):
            return False
        if sd.get("A") and sd["K"] > 0:
            K += sd["K"]
            if M * sd["lda"] >= (1 << 29) or N * sd["ldb"] >= (1 << 29):
                return False
    if len(segs) > 4:  # KS_SEGS (csrc/gemm_kslice.hip)
        return False
    tiles = ((M + 31) // 32) * ((N + 31) // 32)
    return M <= 512 and 128 <= tiles <= 512 and K >= 512


def skinny_n_eligible(amode, bmode, cmode, segs, zmode):
    """mirror of csrc/gemm_skinny.hip `gemm_skinny_n_eligible`: a forward Linear with <= 16 outputs at large batch streams x once, K split
    inside the workgroup (no split-K workspace, no second launch)"""
    if amode != L.AM_KC or bmode not in (L.AM_KC, L.AM_RC) or cmode != L.CM_PLAIN or (zmode and len(segs) != 1):
        return False
    M, N = segs[0]["M"], segs[0]["N"]
    if N < 1 or N > 16 or M < 1024:
        return False
    K = 0
    for sd in segs:
        if sd.get("Aaux") or sd.get("Baux") or sd.get("ones_col") or (0 < sd.get("Mvalid", M) < M):
            return False
        if sd.get("A") and sd["K"] > 0:
            K += sd["K"]
            if M * sd.get("lda", 0) >= (1 << 29) or (sd["K"] if bmode == L.AM_RC else N) * sd.get("ldb", 0) >= (1 << 29):
                return False
    return K >= 256 and _SKINNY_N


_SKINNY_N = _os.environ.get("NASREC_SKINNY_N", "1") != "0"  # A/B knob: 0 keeps these products on the general template (split-K + second pass)
BALANCED_MIN_SAVING_US = float(_os.environ.get("NASREC_BALANCED_MIN_US", "80.0"))  # (env: A/B knob)


def _balanced_schedule_pays(segs, zmode):
    """gemm_fast's balanced schedule (NASREC_SPLITK_BALANCED, csrc/gemm_fast.hip).  Measured model of the plain schedule: a CU
    holds two 128x128 tiles, the chip runs "rounds" of 512 tiles, a round takes k-tiles x 3.8 us (8 x 1024^2 x 4096: 512 tiles
    in 485 us) however full it is — 256 leftover tiles, one per CU, take as long as 512 (1280 tiles: 1878 us against 1582 us at
    the 1536-tile rate) — except that a handful of leftover tiles runs in about half a round (528 tiles: 736 us).  Sharing the
    k-iterations equally costs tiles / 512 rounds plus a second pass of ~35 us (<= 1024 partial tiles of 64 KB out and back):
    take it when the model saves clearly more than that.  Needs the same K in every problem of a batch."""
    live = [sd for sd in segs if sd.get("A")]
    if not live or len(live) != len(segs):
        return False
    if zmode:
        kts = {(sd["K"] + 31) // 32 for sd in segs}
        if len(kts) != 1:
            return False
        T = kts.pop()
        probs = segs
    else:
        T = sum((sd["K"] + 31) // 32 for sd in segs)
        probs = segs[:1]
    tiles = sum(((sd["M"] + 127) // 128) * ((sd["N"] + 127) // 128) for sd in probs)
    left = tiles % 512
    if left == 0:
        return False
    rounds = tiles // 512 + (1.0 if left > 64 else 0.55)
    return (rounds - tiles / 512.0) * T * 3.8 >= BALANCED_MIN_SAVING_US


_TINYK = _os.environ.get("NASREC_TINYK", "1") != "0"  # A/B knob (csrc/gemm_skinny.hip reads the same variable)
_TINYK_RC = _os.environ.get("NASREC_TINYK", "1") == "2"  # the input-gradient form too (measured no faster than the throughput tile: off)


def tinyk_eligible(d) -> bool:
    """mirror of csrc/gemm_skinny.hip `gemm_tinyk_eligible`: K <= 16 at large batch with a wide output (a streaming write)"""
    if not _TINYK or d.amode != L.AM_KC or d.bmode not in (L.AM_KC, L.AM_RC) or d.cmode != L.CM_PLAIN or d.splitk > 1:
        return False
    if d.bmode == L.AM_RC and not _TINYK_RC:
        return False
    if not d.zmode and d.nseg != 1:
        return False
    for q in range(d.nseg):
        s = d.seg[q]
        if s.Aaux or s.Baux or s.ones_col or (0 < s.Mvalid < s.M):
            return False
        if s.M < 1024 or s.N < 256 or s.K < 0 or s.K > 16:
            return False
    return True


def gemm_kernel_name(d) -> str:
    """which kernel family launch_gemm picks for this descriptor (mirror of csrc/gemm.hip / gemm_fast.hip)"""
    segs = [dict(A=d.seg[q].A, Aaux=d.seg[q].Aaux, Baux=d.seg[q].Baux, M=d.seg[q].M, N=d.seg[q].N, K=d.seg[q].K) for q in range(d.nseg)]
    live = [sd for sd in segs if sd["A"]]
    if d.splitk <= 1 and kslice_eligible(d.amode, d.bmode, d.cmode, [dict(sd, ones_col=d.seg[q].ones_col, Mvalid=d.seg[q].Mvalid, lda=d.seg[q].lda,
                                                                          ldb=d.seg[q].ldb) for q, sd in enumerate(segs)], d.zmode):
        return "gemm_kslice_kernel"
    if d.splitk <= 1 and skinny_n_eligible(d.amode, d.bmode, d.cmode, [dict(sd, ones_col=d.seg[q].ones_col, Mvalid=d.seg[q].Mvalid, lda=d.seg[q].lda,
                                                                            ldb=d.seg[q].ldb) for q, sd in enumerate(segs)], d.zmode):
        return "gemm_skinny_n_kernel"
    if tinyk_eligible(d):
        return "gemm_tinyk_kernel"
    if (d.cmode == L.CM_TOKJ and d.bmode == L.AM_TOKR and d.amode in (L.AM_KC, L.AM_RC) and d.splitk <= 1 and not d.pre_add
            and not d.save_act and d.mul_nseg == 0 and all(sd["M"] <= 80 and sd["N"] >= 16 * 1024 and not sd["Aaux"] and not sd["Baux"] for sd in segs)):
        return "token_linear_kernel"  # (csrc/token_linear.hip `token_linear_eligible` has the complete rule)
    if d.cmode != L.CM_PLAIN or (d.amode, d.bmode) not in ((L.AM_KC, L.AM_KC), (L.AM_KC, L.AM_RC), (L.AM_RC, L.AM_RC)) or not live:
        return "gemm_kernel"
    if any(sd["Aaux"] or sd["Baux"] for sd in segs) or max(sd["K"] for sd in live) < _FAST_MIN_K:
        return "gemm_kernel"
    probs = segs if d.zmode else segs[:1]
    tiles = sum(((sd["M"] + 127) // 128) * ((sd["N"] + 127) // 128) for sd in probs)
    if 2 * sum(sd["M"] * sd["N"] for sd in probs) < tiles * 128 * 128:
        return "gemm_kernel"
    return "gemm_fast_kernel" if tiles * max(1, d.splitk) >= GEMM_FAST_MIN_TILES else "gemm_kernel"


def gemm_descs(ctx, amode, bmode, cmode, segs, zmode, act=0, bias_on_rows=0, mask_on_rows=0, dims=-1, beta=0, bias=None, save_z=None,
               save_act=None, pre_add=None, mul=None, splitk=None, rowsum_out=None):
    """One GEMM launch descriptor for <= MAX_SEGS segments (list of dicts with nasrec_gemm_seg_t fields).
    (This function runs ~60 times per compiled supernet path, i.e. per training step: plain loops, no generator expressions.)"""
    if not segs:
        return []
    nseg = len(segs)
    assert nseg <= L.MAX_SEGS
    d = L.GemmDesc()
    d.kind = L.OP_GEMM
    d.amode, d.bmode, d.cmode = amode, bmode, cmode
    d.nseg = nseg
    d.zmode = zmode
    if act:
        d.act = act
    if bias_on_rows:
        d.bias_on_rows = bias_on_rows
    if mask_on_rows:
        d.mask_on_rows = mask_on_rows
    d.dims_in_use = dims
    if beta:
        d.beta = beta
    if bias:
        d.bias = bias
    if save_z:
        d.save_z = save_z
    if save_act:
        d.save_act = save_act
    if pre_add:
        d.pre_add = pre_add
    if mul:
        assert len(mul) <= L.MAX_SEGS
        d.mul_nseg = len(mul)
        for q, (ptr, off, width, ld) in enumerate(mul):
            d.mul_ptr[q], d.mul_off[q], d.mul_width[q], d.mul_ld[q] = ptr, off, width, ld
    M = N = 0
    kt_max = kt_sum = live_tiles = 0
    dseg = d.seg
    for q in range(nseg):
        sdict = segs[q]
        s = dseg[q]
        for k, v in sdict.items():
            setattr(s, k, v)
        m, n = sdict["M"], sdict["N"]
        if "Mvalid" not in sdict:
            s.Mvalid = m
        if m > M:
            M = m
        if n > N:
            N = n
        kt = (sdict["K"] + 31) // 32
        kt_sum += kt
        if kt > kt_max:
            kt_max = kt
        live_tiles += ((m + 63) // 64) * ((n + 63) // 64)  # (a zmode grid is padded to Mmax x Nmax)
    B = ctx.B
    if zmode:
        S = _splitk_for(live_tiles, kt_max)
    else:
        S = _splitk_for(((M + 63) // 64) * ((N + 63) // 64), kt_sum)
    fast = _fast_gemm_splitk(amode, bmode, cmode, segs, zmode) if B > 128 else None  # (a throughput-regime launch needs >= 120 128x128 tiles)
    if fast is not None:
        S = fast
    if amode == L.AM_TOKK and zmode and B >= 1024:
        ok = True
        for sd in segs:
            if sd["M"] > 80 or sd["N"] > 80 or sd.get("Aaux") or sd.get("Baux"):
                ok = False
                break
        if ok:
            # token-axis weight gradients at large batch (csrc/token_linear.hip `token_dw_kernel`): S workgroups of 16 wavefronts per
            # problem, a wavefront per sample — one workgroup per CU whatever the number of problems, few slabs for the second pass
            S = max(4, min(_TOKDW_CAP, 256 // nseg, B // 64))
    if splitk is not None:
        S = splitk
    elif B <= 512 and S > 1 and kslice_eligible(amode, bmode, cmode, segs, zmode):
        S = 1  # csrc/gemm_kslice.hip: K is split inside the workgroup
    elif B >= 1024 and skinny_n_eligible(amode, bmode, cmode, segs, zmode):
        S = 1  # csrc/gemm_skinny.hip: likewise
    d.splitk = 1
    if fast == 1 and S == 1 and B > 256 and ctx.sk_workspace is not None and _balanced_schedule_pays(segs, zmode):
        d.splitk = L.SPLITK_BALANCED
        d.workspace = ctx.sk_workspace().data_ptr()
    if S > 1:
        d.splitk = S
        ws = ctx.alloc(S * M * N * (nseg if zmode else 1))
        d.workspace = ws.data_ptr()
    if rowsum_out is not None:
        d.rowsum_out = rowsum_out
    return [d]


# ----------------------------------------------------------------------------------------------------------------
# emitters (forward descriptor(s) + backward closure)
# ----------------------------------------------------------------------------------------------------------------
def _live_segs(segs: List[Seg]):
    return [s for s in segs if s.view is not None]


def _cover_pieces(live: List[Seg]):
    """Split the K-axis footprints of the segments into disjoint pieces.  Returns [(seg, a, b, rank)]: columns
    [a, b) of the consumer's K axis are covered by `seg` as the rank-th writer.  Sum feeds two segment lists into the
    same K range (modules.py:487), so dW[:, a:b] receives one product per covering segment: rank 0 overwrites, later
    ranks accumulate in later launches."""
    cuts = sorted({s.koff for s in live} | {s.koff + s.width for s in live})
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rank = 0
        for s in live:
            if s.koff <= a and b <= s.koff + s.width:
                if out and out[-1][0] is s and out[-1][2] == a and out[-1][3] == rank:
                    out[-1] = (s, out[-1][1], b, rank)
                else:
                    out.append((s, a, b, rank))
                rank += 1
    return out


def _split_column_remainder(ctx, bm, cm, d):
    """[problem] or [its first 128 n columns, its last <= 16 columns].  A [B, 1035] input gradient or a [1024, 1035 (+1)] weight
    gradient (13 dense features + 1024) has a ninth column of 128-wide tiles that is 8 % full: 576 tiles instead of 512 put a
    second, almost empty round on the chip (8192 x 1035 x 1024: 200 us against 123 us for 8192 x 1024 x 1024).  The remainder
    columns become a problem of their own, which the shape classes send to the small-tile kernel."""
    if ctx.B <= 256 or cm != L.CM_PLAIN or bm not in (L.AM_KC, L.AM_RC) or not d.get("A"):
        return [d]
    N, ones = d["N"], int(bool(d.get("ones_col")))
    r = N % 128
    if N <= 128 or r == 0 or r > 16 or r - ones < 1 or d["M"] < 128:
        return [d]
    Nm = N - r
    step = 4 * Nm * (d["ldb"] if bm == L.AM_KC else 1)
    main = dict(d, N=Nm, ones_col=0, rowsum=None)
    rem = dict(d, N=r, B=d["B"] + step, C=d["C"] + 4 * Nm)
    if d.get("Baux"):
        rem["Baux"] = d["Baux"] + step
    return [main, rem]


def _emit_z_groups(ctx, am, bm, cm, items, rowsum_out=None):
    """items: [(group_index, seg_dict)] -> one launch per group (in order), <= MAX_SEGS problems per launch.
    rowsum_out: fuse the bias gradient into the first problem as a virtual ones-column (B(N-1,k) = 1)."""
    items = [(g, p) for g, d in _with_rowsum(items, rowsum_out) for p in _split_column_remainder(ctx, bm, cm, d)]
    for gi in sorted({g for g, _ in items}):
        grp = [d for g, d in items if g == gi]
        for cls in sorted({_shape_class(ctx, d) for d in grp}):
            part = [d for d in grp if _shape_class(ctx, d) == cls]
            for i in range(0, len(part), L.MAX_SEGS):
                for d in gemm_descs(ctx, am, bm, cm, part[i:i + L.MAX_SEGS], 1):
                    ctx.emit(d)


def _shape_class(ctx, d):
    """problems of one zmode launch share its tile shape (chosen from Mmax x Nmax): in the throughput regime skinny ones get
    launches of their own (at batch <= 256 every launch is latency-bound and fewer launches win: 0.61 vs 0.67 ms per step)"""
    return (d["M"] < 64, d["N"] < 64) if ctx.B > 256 else (False, False)


def _with_rowsum(items, rowsum_out):
    if rowsum_out is None or not items:
        return list(items)
    items = list(items)
    g0 = min(g for g, _ in items)
    i0 = next(i for i, (g, _) in enumerate(items) if g == g0)
    g, d = items[i0]
    items[i0] = (g, dict(d, N=d["N"] + 1, ones_col=1, rowsum=rowsum_out))
    return items


def _weight_grad_products(ctx, am, bm, cm, items, rowsum_out=None):
    """dW products have no consumer inside the backward pass (only the optimizer / the gradient exchange read them), and
    their operands (dz, the saved activations) stay untouched once produced.  At batch 256 each of them is a
    launch-latency-bound kernel (+ a split-K second pass), so they are parked and issued at the END of the backward
    program as a few zmode launches of up to MAX_SEGS independent problems each (_flush_deferred)."""
    if not ctx.defer_dw:
        return _emit_z_groups(ctx, am, bm, cm, items, rowsum_out)
    for rank, d in _with_rowsum(items, rowsum_out):
        for p in _split_column_remainder(ctx, bm, cm, d):
            ctx.deferred.append((am, bm, cm, rank, p))


def _flush_mha_reduce(ctx):
    """one partial buffer [B, n * MHA_PARAMS] for the n Transformer nodes of the step, one NASREC_OP_REDUCE_ROWS over it"""
    jobs, ctx.mha_reduce = ctx.mha_reduce, []
    per = L.REDUCE_MAX_DST // 12
    for i0 in range(0, len(jobs), per):
        grp = jobs[i0:i0 + per]
        ld = len(grp) * L.MHA_PARAMS
        part = ctx.alloc(ctx.B * ld)
        r = L.ReduceRowsDesc()
        r.kind = L.OP_REDUCE_ROWS
        r.R, r.C, r.ld = ctx.B, ld, ld
        r.in_ = part.data_ptr()
        n = 0
        for j, (e, dsts) in enumerate(grp):
            e.dparams_partial = part.data_ptr() + 4 * j * L.MHA_PARAMS
            e.partial_ld = ld
            for ptr, off, length in dsts:
                r.dst[n], r.dst_off[n], r.dst_len[n] = ptr, j * L.MHA_PARAMS + off, length
                n += 1
        r.ndst = n
        ctx.emit(r)


_LN_REDUCE_BATCH = _os.environ.get("NASREC_LN_REDUCE_BATCH", "1") != "0"


class _ItemNode:
    """what reports (bench.py, iter_ops) read of a worklist item — descriptor, part, footprints — with the footprints computed when
    somebody asks (schedule.desc_io costs ~10 us per descriptor: not on the path of a per-step compile)"""
    __slots__ = ("desc", "part", "_io", "level", "index")

    def __init__(self, desc, part="whole"):
        self.desc, self.part, self._io, self.level, self.index = desc, part, None, 0, 0

    def _get(self):
        if self._io is None:
            from . import schedule as S
            io = S.desc_io(self.desc, self.part)
            self._io = io if io is not None else (None, None)
        return self._io

    @property
    def reads(self):
        return self._get()[0]

    @property
    def writes(self):
        return self._get()[1]


def _flush_ln_reduce(ctx):
    """the parked LayerNorm parameter-gradient reductions (fixed-order column sums of per-workgroup partials) as items of
    NASREC_OP_WORKLIST launches, twelve per launch: same body, same order of additions, one launch boundary instead of twelve"""
    jobs, ctx.ln_reduce = ctx.ln_reduce, []
    if not jobs:
        return
    size = (C.sizeof(L.WlReduce) + 15) & ~15
    base = L.WorklistDesc.blob.offset
    for i0 in range(0, len(jobs), L.WL_MAX_ITEMS):
        grp = jobs[i0:i0 + L.WL_MAX_ITEMS]
        if len(grp) == 1:
            ctx.emit(grp[0])
            continue
        w = L.WorklistDesc()
        w.kind, w.n = L.OP_WORKLIST, len(grp)
        w.nodes = [_ItemNode(r) for r in grp]  # (for reports: footprints are derived on demand, not on the per-step path)
        assert len(grp) * size <= L.WL_BLOB_BYTES
        for k, r in enumerate(grp):
            it = w.item[k]
            it.kind, it.part, it.off = r.kind, L.WL_WHOLE, k * size
            item = L.WlReduce.from_address(C.addressof(w) + base + k * size)  # the item's descriptor, written in place
            item.kind, item.R, item.C, item.ld, item.in_, item.ndst = r.kind, r.R, r.C, r.ld, r.in_, r.ndst
            for q in range(r.ndst):
                item.dst[q], item.dst_off[q], item.dst_len[q] = r.dst[q], r.dst_off[q], r.dst_len[q]
        ctx.emit(w)


def _flush_deferred(ctx, todo=None):
    final = todo is None  # the flush at the end of the backward program (not an early flush of a few products)
    if todo is None:
        todo, ctx.deferred = ctx.deferred, []
    keys = sorted({(rank, am, bm, cm, d["K"], _shape_class(ctx, d)) for am, bm, cm, rank, d in todo})  # rank r accumulates over rank r-1: later launch
    split = []

    def second_passes():
        # Batch 256: the parked launches have no consumers before the optimizer, so their split-K second passes need not follow
        # them one by one — up to three run as ONE launch behind the last of them (each launch on the critical path costs ~5 us).
        # Rank-0 products only, and before any rank-1 product (which accumulates over what a rank-0 second pass wrote).
        if len(split) > 1:
            for g in split:
                g.defer_second_pass = 1
            for i in range(0, len(split), L.EPILOGUES_MAX):
                chunk = split[i:i + L.EPILOGUES_MAX]
                e = L.SplitkEpiloguesDesc()
                e.kind, e.n = L.OP_SPLITK_EPILOGUES, len(chunk)
                for q, g in enumerate(chunk):
                    C.memmove(C.addressof(e.g[q]), C.addressof(g), C.sizeof(L.GemmDesc))
                ctx.emit(e)
        del split[:]

    for key in keys:
        if key[0] > 0:
            second_passes()
        grp = [d for am, bm, cm, rank, d in todo if (rank, am, bm, cm, d["K"], _shape_class(ctx, d)) == key]
        grp.sort(key=lambda d: -d["M"] * d["N"])  # problems of similar size share a launch (its grid is Mmax x Nmax)
        for i in range(0, len(grp), L.MAX_SEGS):
            for g in gemm_descs(ctx, key[1], key[2], key[3], grp[i:i + L.MAX_SEGS], 1):
                ctx.emit(g)
                if final and ctx.B <= 256 and g.splitk > 1 and g.cmode == L.CM_PLAIN and key[0] == 0 and not ctx.shape_only:
                    split.append(g)
    second_passes()


def _flush_raw_dx(ctx):
    """Gradients into the raw embedding tokens have no consumer inside the backward (the stem's output is a leaf: only the
    row-sparse optimizer reads its gradient), and every block's token-axis Linear that reads the raw tokens contributes
    dx = W_l^T dz_l to the SAME [B, Fs, 16] buffer.  Instead of one accumulating launch per contributor, in backward order, they are
    parked and issued as ONE K-concatenated product [W_1^T | W_2^T | ...] [dz_1; dz_2; ...] at the end of the backward chain
    (batch 256: five 5.6-10.8 us launches -> one)."""
    items, ctx.deferred_raw = ctx.deferred_raw, []
    if not items:
        return
    view = items[0]["view"]
    gp, acc = ctx.gtarget(view)
    if gp is None:
        return
    for i in range(0, len(items), L.MAX_SEGS):
        segs = [dict(A=e["A"], B=e["B"], Baux=e["Baux"], C=gp, M=e["M"], N=e["N"], K=e["K"], lda=e["lda"], ldb=e["ldb"], ldc=view.ld)
                for e in items[i:i + L.MAX_SEGS]]
        for d in gemm_descs(ctx, L.AM_RC, L.AM_TOKR, L.CM_TOKJ, segs, 0, beta=int(acc or i > 0)):
            ctx.emit(d)


def _is_whole_raw(ctx, s):
    """segment = all tokens of the embedding stem's output?"""
    v = s.view
    return ctx.defer_dw and ctx.raw_sparse is not None and v.buf is ctx.raw_sparse and v.off == 0 and v.N * E == v.ld and s.width == v.N


def _dx_groups(ctx, live, mk):
    """gradient products towards the input segments; two segments that alias the same buffer (Sum/gating with
    left == right) must not race inside one launch -> they go to consecutive launches."""
    items, seen = [], {}
    for s in live:
        gp, acc = ctx.gtarget(s.view)
        if gp is None:
            continue
        g = seen.get(id(s.view.buf), 0)
        seen[id(s.view.buf)] = g + 1
        items.append((g, mk(s, gp, acc)))
    return items


def emit_layernorm(ctx, mode, x_ptr, ldx, R, D, wname, out_ptr, ldy, act, dims, accumulate, out_view, x_grad_cb):
    """LN forward + closure. x_grad_cb(dx_tensor) is called in backward with the tensor holding dL/dx
    (same layout as x) and must emit the producer's backward."""
    w = ctx.param(wname + ".weight", (D,))
    b = ctx.param(wname + ".bias", (D,))
    if ctx.shape_only:
        return
    stats = ctx.alloc(R * 2)
    d = L.LayerNormDesc()
    d.kind = L.OP_LAYERNORM_FWD
    d.mode, d.R, d.D, d.ldx, d.ldy = mode, R, D, ldx, ldy
    d.act, d.dims_in_use, d.accumulate, d.eps = act, dims, accumulate, 1e-5
    d.x, d.w, d.b, d.y, d.stats = x_ptr, w, b, out_ptr, stats.data_ptr()
    ctx.emit(d)

    def bwd():
        if not ctx.live(out_view):
            return
        nblk = min((R + 3) // 4, _LN_NBLK) if mode == L.AM_KC else (R + 255) // 256
        part = ctx.alloc(nblk * 2 * D)
        dx = ctx.alloc(_ln_numel(mode, R, D, ldx))
        e = L.LayerNormDesc()
        e.kind = L.OP_LAYERNORM_BWD
        e.mode, e.R, e.D, e.ldx, e.ldy = mode, R, D, ldx, ldy
        e.act, e.dims_in_use, e.accumulate, e.eps = act, dims, 0, 1e-5
        e.x, e.w, e.b, e.stats = x_ptr, w, b, stats.data_ptr()
        e.dy = out_view.gptr
        e.dx = dx.data_ptr()
        e.dwb_partial = part.data_ptr()
        e.nblk = nblk
        ctx.emit(e)
        r = L.ReduceRowsDesc()
        r.kind = L.OP_REDUCE_ROWS
        r.R, r.C, r.ld = nblk, 2 * D, 2 * D
        r.in_ = part.data_ptr()
        r.ndst = 2
        r.dst[0], r.dst_off[0], r.dst_len[0] = ctx.gparam(wname + ".weight"), 0, D
        r.dst[1], r.dst_off[1], r.dst_len[1] = ctx.gparam(wname + ".bias"), D, D
        if ctx.B > 256 and _LN_REDUCE_BATCH:
            # large batch: ~20 of these per step, 6 us each and a launch boundary each, with no consumer before the optimizer — they
            # wait for the end of the block (data parallel) / of the backward and go out as items of ONE heterogeneous launch
            ctx.ln_reduce.append(r)
        else:
            ctx.emit(r)
        x_grad_cb(dx)

    ctx.on_backward(bwd)


def _ln_numel(mode, R, D, ld):
    if mode == L.AM_KC:
        return R * ld
    return (R // 16) * ld


def linear_dense(ctx, segs: List[Seg], Ktot, wname, nout, bias: bool, out: DV, act=L.ACT_NONE, dims=-1, beta=0, ln=None,
                 mul=None, save_act: Optional[DV] = None, pre_dz_cb=None, pre_add: Optional[DV] = None):
    """y = epilogue(concat(segs) · Wᵀ + b).  `ln`: parameter prefix of a LayerNorm applied after the product
    (then act/mask/beta move to the LN kernel).  `mul`/`save_act`: SigmoidGating epilogue.  Returns nothing;
    registers the backward closure.  pre_dz_cb: if given, called in backward to obtain (dz_ptr, ld) instead of
    using out.grad (gating).  pre_add: a [B, nout] view with out's row stride whose value is added to the product before
    bias/activation (out = pre_add + x Wᵀ + b written in one pass instead of copy + accumulate); its gradient is the
    caller's business."""
    B = ctx.B
    W = ctx.param(wname + ".weight", (nout, Ktot))
    bptr = ctx.param(wname + ".bias", (nout,)) if bias else None
    if ctx.shape_only:
        if ln:
            ctx.param(ln + ".weight", (nout,))
            ctx.param(ln + ".bias", (nout,))
        return
    live = _live_segs(segs)
    if pre_add is not None:
        assert not ln and act == L.ACT_NONE and dims < 0 and beta == 0 and pre_add.ld == out.ld and pre_add.width == nout
    if len(live) > L.MAX_SEGS and not ln and (act != L.ACT_NONE or mul is not None):
        raise NotImplementedError("more than %d live input segments for the activated product %s" % (L.MAX_SEGS, wname))
    need_z = act == L.ACT_SILU and not ln  # SiLU backward needs the pre-activation
    zbuf = None
    if ln:
        zbuf = ctx.alloc(B * nout)
        tgt_ptr, tgt_ld = zbuf.data_ptr(), nout
        g_act, g_dims, g_beta = L.ACT_NONE, -1, 0
    else:
        tgt_ptr, tgt_ld = out.ptr, out.ld
        g_act, g_dims, g_beta = act, dims, beta
    savez = ctx.alloc(B * nout) if need_z else None
    sd = [dict(A=s.view.ptr, B=W + 4 * s.koff, C=tgt_ptr, M=B, N=nout, K=s.width, lda=s.view.ld, ldb=Ktot, ldc=tgt_ld) for s in live]
    if not sd:  # every input is a zero segment: the product is 0
        sd = [dict(A=None, B=W, C=tgt_ptr, M=B, N=nout, K=0, lda=1, ldb=Ktot, ldc=tgt_ld)]
    mul_list = None
    if mul is not None:
        mul_list = [(m.view.ptr, m.koff, m.width, m.view.ld) for m in mul if m.view is not None]
        if not mul_list:
            mul_list = [(None, 0, 1, 1)]
    # a K range of more than MAX_SEGS live segments (Sum on the full path) is a sum of products: chain with beta
    for c0 in range(0, len(sd), L.MAX_SEGS):
        for d in gemm_descs(ctx, L.AM_KC, L.AM_KC, L.CM_PLAIN, sd[c0:c0 + L.MAX_SEGS], 0, act=g_act, dims=g_dims,
                            beta=g_beta if c0 == 0 else 1, bias=bptr if c0 == 0 else None,
                            save_z=savez.data_ptr() if need_z else None, save_act=save_act.ptr if save_act is not None else None,
                            mul=mul_list, pre_add=pre_add.ptr if (pre_add is not None and c0 == 0) else None):
            ctx.emit(d)

    def backward_products(dz_ptr, dz_ld, aux_ptr, kdims):
        """dz: [B, nout] with row stride dz_ld; aux: optional ReLU mask source (same layout); kdims: number of
        leading output columns that can carry gradient (prefix mask)."""
        kd = nout if kdims < 0 else min(kdims, nout)
        # dX per live segment that wants a gradient
        _emit_z_groups(ctx, L.AM_KC, L.AM_RC, L.CM_PLAIN, _dx_groups(ctx, live, lambda s, gp, acc: dict(
            A=dz_ptr, Aaux=aux_ptr, B=W + 4 * s.koff, C=gp, M=B, N=s.width, K=kd, lda=dz_ld, ldb=Ktot, ldc=s.view.ld, accumulate=acc)))
        # dW per covered piece of the K axis (columns of zero segments keep the zero the flat gradient buffer was reset to)
        gW = ctx.gparam(wname + ".weight")
        _weight_grad_products(ctx, L.AM_RC, L.AM_RC, L.CM_PLAIN, [(rank, dict(
            A=dz_ptr, Aaux=aux_ptr, B=s.view.ptr + 4 * (a - s.koff), C=gW + 4 * a, M=nout, N=b - a, K=B, lda=dz_ld, ldb=s.view.ld, ldc=Ktot,
            Mvalid=kd, accumulate=int(rank > 0))) for s, a, b, rank in _cover_pieces(live)],
            rowsum_out=ctx.gparam(wname + ".bias") if (bias and live) else None)
        if bias and not live:
            r = L.RowsumDesc()
            r.kind = L.OP_ROWSUM
            r.mode, r.R, r.K, r.ld, r.rvalid = L.AM_RC, nout, B, dz_ld, kd
            r.p, r.aux, r.out = dz_ptr, aux_ptr, ctx.gparam(wname + ".bias")
            ctx.emit(r)

    if ln:
        emit_layernorm(ctx, L.AM_KC, zbuf.data_ptr(), nout, B, nout, ln, out.ptr, out.ld, act, dims, beta, out,
                       lambda dx: backward_products(dx.data_ptr(), nout, None, -1))
        return

    def bwd():
        if pre_dz_cb is not None:
            r = pre_dz_cb()
            if r is None:
                return
            dz_ptr, dz_ld = r
            backward_products(dz_ptr, dz_ld, None, -1)
            return
        if not ctx.live(out):
            return
        if act == L.ACT_RELU:
            backward_products(out.gptr, out.ld, out.ptr, dims)
        elif act == L.ACT_SILU:
            dz = ctx.alloc(B * nout)
            a = L.ActBwdDesc()
            a.kind = L.OP_ACT_BWD
            a.mode, a.R, a.D = L.AM_KC, B, nout
            a.ld_dy, a.ld_z, a.ld_dz = out.ld, nout, nout
            a.act, a.dims_in_use = act, dims
            a.dy, a.z, a.dz = out.gptr, savez.data_ptr(), dz.data_ptr()
            ctx.emit(a)
            backward_products(dz.data_ptr(), nout, None, -1)
        else:
            backward_products(out.gptr, out.ld, None, dims)

    ctx.on_backward(bwd)


def linear_tokens(ctx, segs: List[Seg], Ntot, wname, nout, bias: bool, out: SV, act=L.ACT_NONE, dims=-1, ln=None):
    """Token-axis Linear (modules.py:222-234, 358-361, 648-650): out[b,n',e] = sum_n W[n',n] x[b,n,e] (+ b[n']),
    optional LayerNorm over n', activation, prefix mask over n'."""
    B = ctx.B
    W = ctx.param(wname + ".weight", (nout, Ntot))
    bptr = ctx.param(wname + ".bias", (nout,)) if bias else None
    if ctx.shape_only:
        if ln:
            ctx.param(ln + ".weight", (nout,))
            ctx.param(ln + ".bias", (nout,))
        return
    live = _live_segs(segs)
    if len(live) > L.MAX_SEGS:
        raise NotImplementedError("more than %d live input segments for %s" % (L.MAX_SEGS, wname))
    need_z = act == L.ACT_SILU and not ln
    if ln:
        zbuf = ctx.alloc(B * nout * E)
        tgt_ptr, tgt_ld = zbuf.data_ptr(), nout * E
        g_act, g_dims = L.ACT_NONE, -1
    else:
        tgt_ptr, tgt_ld = out.ptr, out.ld
        g_act, g_dims = act, dims
    savez = ctx.alloc(B * nout * E) if need_z else None
    sd = [dict(A=W + 4 * s.koff, B=s.view.ptr, C=tgt_ptr, M=nout, N=B * E, K=s.width, lda=Ntot, ldb=s.view.ld, ldc=tgt_ld) for s in live]
    if not sd:
        sd = [dict(A=None, B=W, C=tgt_ptr, M=nout, N=B * E, K=0, lda=Ntot, ldb=1, ldc=tgt_ld)]
    for d in gemm_descs(ctx, L.AM_KC, L.AM_TOKR, L.CM_TOKJ, sd, 0, act=g_act, dims=g_dims, bias=bptr, bias_on_rows=1, mask_on_rows=1,
                        save_z=savez.data_ptr() if need_z else None):
        ctx.emit(d)

    def backward_products(dz_ptr, dz_ld, aux_ptr, kdims):
        kd = nout if kdims < 0 else min(kdims, nout)
        for s in live:
            if _is_whole_raw(ctx, s) and s.view.buf.need_grad:  # parked: one K-concatenated launch for all of them (_flush_raw_dx)
                ctx.deferred_raw.append(dict(A=W + 4 * s.koff, B=dz_ptr, Baux=aux_ptr, M=s.width, N=B * E, K=kd, lda=Ntot, ldb=dz_ld, view=s.view))
        rest = [s for s in live if not (_is_whole_raw(ctx, s) and s.view.buf.need_grad)]
        _emit_z_groups(ctx, L.AM_RC, L.AM_TOKR, L.CM_TOKJ, _dx_groups(ctx, rest, lambda s, gp, acc: dict(
            A=W + 4 * s.koff, B=dz_ptr, Baux=aux_ptr, C=gp, M=s.width, N=B * E, K=kd, lda=Ntot, ldb=dz_ld, ldc=s.view.ld, accumulate=acc)))
        gW = ctx.gparam(wname + ".weight")
        _weight_grad_products(ctx, L.AM_TOKK, L.AM_TOKK, L.CM_PLAIN, [(rank, dict(
            A=dz_ptr, Aaux=aux_ptr, B=s.view.ptr + 4 * (a - s.koff) * E, C=gW + 4 * a, M=nout, N=b - a, K=B * E, lda=dz_ld, ldb=s.view.ld,
            ldc=Ntot, Mvalid=kd, accumulate=int(rank > 0))) for s, a, b, rank in _cover_pieces(live)],
            rowsum_out=ctx.gparam(wname + ".bias") if (bias and live) else None)
        if bias and not live:
            r = L.RowsumDesc()
            r.kind = L.OP_ROWSUM
            r.mode, r.R, r.K, r.ld, r.rvalid = L.AM_TOKK, nout, B * E, dz_ld, kd
            r.p, r.aux, r.out = dz_ptr, aux_ptr, ctx.gparam(wname + ".bias")
            ctx.emit(r)

    if ln:
        emit_layernorm(ctx, L.AM_TOKR, zbuf.data_ptr(), nout * E, B * E, nout, ln, out.ptr, out.ld, act, dims, 0, out,
                       lambda dx: backward_products(dx.data_ptr(), nout * E, None, -1))
        return

    def bwd():
        if not ctx.live(out):
            return
        if act == L.ACT_RELU:
            backward_products(out.gptr, out.ld, out.ptr, dims)
        elif act == L.ACT_SILU:
            dz = ctx.alloc(B * nout * E)
            a = L.ActBwdDesc()
            a.kind = L.OP_ACT_BWD
            a.mode, a.R, a.D = L.AM_TOKR, B * E, nout
            a.ld_dy, a.ld_z, a.ld_dz = out.ld, nout * E, nout * E
            a.act, a.dims_in_use = act, dims
            a.dy, a.z, a.dz = out.gptr, savez.data_ptr(), dz.data_ptr()
            ctx.emit(a)
            backward_products(dz.data_ptr(), nout * E, None, -1)
        else:
            backward_products(out.gptr, out.ld, None, dims)

    ctx.on_backward(bwd)


def copy_into(ctx, segs: List[Seg], dst: DV, accumulate=0):
    """dst[b, koff+j] (+)= seg[b, j]; backward: seg.grad += dst.grad slice."""
    if ctx.shape_only:
        return
    live = _live_segs(segs)
    for i in range(0, max(len(live), 1), L.MAX_SEGS):
        chunk = live[i:i + L.MAX_SEGS]
        d = L.CopySegsDesc()
        d.kind = L.OP_COPY_SEGS
        d.B, d.nseg, d.ld_dst, d.accumulate, d.reverse = ctx.B, len(chunk), dst.ld, accumulate, 0
        d.dst = dst.ptr
        for q, s in enumerate(chunk):
            d.seg[q], d.width[q], d.ld[q], d.off[q] = s.view.ptr, s.width, s.view.ld, s.koff
        if chunk:
            ctx.emit(d)

    def bwd():
        if not ctx.live(dst):
            return
        for i in range(0, len(live), L.MAX_SEGS):
            chunk = live[i:i + L.MAX_SEGS]
            d = L.CopySegsDesc()
            d.kind = L.OP_COPY_SEGS
            d.B, d.ld_dst, d.reverse = ctx.B, dst.ld, 1
            d.dst = dst.gptr
            n = 0
            for s in chunk:
                gp, acc = ctx.gtarget(s.view)
                if gp is None:
                    continue
                d.seg[n], d.width[n], d.ld[n], d.off[n], d.seg_accumulate[n] = gp, s.width, s.view.ld, s.koff, acc
                n += 1
            d.nseg = n
            if n:
                ctx.emit(d)

    ctx.on_backward(bwd)


def _clip_segs(segs: List[Seg], mask: int):
    """A 0/1 prefix mask over a virtual concatenation (modules.py:57-96) == the same concatenation cut at column `mask`."""
    if mask < 0:
        return list(segs)
    out = []
    for s in segs:
        w = min(s.width, max(0, mask - s.koff))
        if w <= 0:
            continue
        v = s.view
        if v is not None and w != s.width:
            v = DV(v.buf, v.off, w, v.ld)
        out.append(Seg(v, s.koff, w))
    return out


def masked_copy(ctx, src: DV, mask: int, tgt: "Target"):
    """tgt (+)= src * prefix_mask(mask) for an operator that skipped its projection and has no LayerNorm (the mask is then
    the only thing left between the operator's core and the node sum)."""
    if not tgt.accumulate:
        zero_fill(ctx, tgt.view)
    copy_into(ctx, _clip_segs([Seg(src, 0, src.width)], mask), tgt.view, 1)


def zero_fill(ctx, view_dense: DV):
    """write zeros into a dense view (copy of a single zero segment)"""
    if ctx.shape_only:
        return
    d = L.CopySegsDesc()
    d.kind = L.OP_COPY_SEGS
    d.B, d.nseg, d.ld_dst, d.accumulate, d.reverse = ctx.B, 1, view_dense.ld, 0, 0
    d.dst = view_dense.ptr
    d.seg[0], d.width[0], d.ld[0], d.off[0] = None, view_dense.width, 1, 0
    ctx.emit(d)


# ----------------------------------------------------------------------------------------------------------------
# operators
# ----------------------------------------------------------------------------------------------------------------
class Target:
    """where a node writes its [B, width] (dense) or [B, N, 16] (sparse) result, and whether it must add to it"""

    def __init__(self, view, accumulate):
        self.view, self.accumulate = view, accumulate


def op_elastic_linear(ctx, cfg, pre, segs, Ktot, max_dims, dims, tgt: Target):
    """ElasticLinear modules.py:134-181"""
    act = L.ACT_BY_NAME[cfg.activation]
    use_ln = cfg.use_layernorm
    mask = -1 if cfg.fixed else dims
    if use_ln:
        linear_dense(ctx, segs, Ktot, pre + "._linear", max_dims, False, tgt.view, act, mask, tgt.accumulate, ln=pre + "._layernorm")
    else:
        assert not (tgt.accumulate and act != L.ACT_NONE), "activated node output needs a private buffer"
        linear_dense(ctx, segs, Ktot, pre + "._linear", max_dims, True, tgt.view, act, mask, tgt.accumulate)


def op_elastic_linear3d(ctx, cfg, pre, segs, Ntot, max_dims, dims, out: SV):
    """ElasticLinear3D modules.py:184-235"""
    act = L.ACT_BY_NAME[cfg.activation]
    mask = -1 if cfg.fixed else dims
    if cfg.use_layernorm:
        linear_tokens(ctx, segs, Ntot, pre + "._linear", max_dims, False, out, act, mask, ln=pre + "._layernorm")
    else:
        linear_tokens(ctx, segs, Ntot, pre + "._linear", max_dims, True, out, act, mask)


def op_dot_product(ctx, cfg, pre, dsegs, Dtot, ssegs, Ntot, max_dims, dims, tgt: Target):
    """DotProduct modules.py:273-401"""
    B = ctx.B
    use_ln = cfg.use_layernorm
    k = round(math.sqrt(2 * max_dims))  # modules.py:298
    k1 = k + 1
    P = k1 * k // 2
    Tbuf = ctx.buf(B * k1 * E)
    Tx = DV(Tbuf, 0, E, k1 * E)  # row 0
    Ty = SV(Tbuf, E, k, k1 * E)  # rows 1..k
    # dense side
    if Dtot != E:
        linear_dense(ctx, dsegs, Dtot, pre + "._dense_proj", E, not use_ln, Tx, L.ACT_NONE, -1, 0,
                     ln=(pre + "._dense_layernorm") if use_ln else None)
    else:
        copy_into(ctx, dsegs, Tx)
    # sparse side (E is always 16, so _sparse_proj is always dropped, modules.py:347-354)
    if Ntot != k:
        linear_tokens(ctx, ssegs, Ntot, pre + "._sparse_inp_proj", k, not use_ln, Ty, L.ACT_NONE, -1,
                      ln=(pre + "._sparse_inp_proj_layernorm") if use_ln else None)
    else:
        copy_into(ctx, [Seg(s.view.dense() if s.view is not None else None, s.koff * E, s.width * E) for s in ssegs], Ty.dense())
    tri = ctx.buf(B * P)
    triv = DV(tri, 0, P, P)
    if not ctx.shape_only:
        d = L.DotTriDesc()
        d.kind = L.OP_DOT_TRI_FWD
        d.B, d.k1, d.ld_out = B, k1, P
        d.T, d.out = Tbuf.t.data_ptr(), tri.t.data_ptr()
        ctx.emit(d)

        def bwd():
            if not ctx.live(triv):
                return
            e = L.DotTriDesc()
            e.kind = L.OP_DOT_TRI_BWD
            e.B, e.k1, e.ld_out = B, k1, P
            e.T, e.dout, e.dT = Tbuf.t.data_ptr(), tri.grad_tensor().data_ptr(), Tbuf.grad_tensor().data_ptr()
            Tbuf.mark()
            ctx.emit(e)

        ctx.on_backward(bwd)
    mask = -1 if cfg.fixed else dims
    if P == max_dims:  # the triangle already has the target width: _linear_proj is dropped, the LayerNorm stays (modules.py:383-392)
        if use_ln:
            emit_ln_dense(ctx, triv, P, pre + "._linear_layernorm", tgt, L.ACT_NONE, mask)
        elif not ctx.shape_only:
            masked_copy(ctx, triv, mask, tgt)
        return
    linear_dense(ctx, [Seg(triv, 0, P)], P, pre + "._linear_proj", max_dims, not use_ln, tgt.view, L.ACT_NONE, mask, tgt.accumulate,
                 ln=(pre + "._linear_layernorm") if use_ln else None)


def _pad_width(lsegs, Ltot, rsegs, Rtot):
    return max(Ltot, Rtot)


def op_sum(ctx, cfg, pre, lsegs, Ltot, rsegs, Rtot, max_dims, dims, tgt: Target):
    """Sum modules.py:432-501 (+ zero padding :403-430).  Linear(left+right) == one GEMM over both segment lists."""
    D = _pad_width(lsegs, Ltot, rsegs, Rtot)
    use_ln = cfg.use_layernorm
    mask = -1 if cfg.fixed else dims
    both = list(lsegs) + list(rsegs)
    if D != max_dims:
        linear_dense(ctx, both, D, pre + "._linear_proj", max_dims, not use_ln, tgt.view, L.ACT_NONE, mask, tgt.accumulate,
                     ln=(pre + "._layernorm") if use_ln else None)
        return
    # no projection: out = left + right (then LN / mask)
    if use_ln:
        tmp = ctx.buf(ctx.B * D)
        tv = DV(tmp, 0, D, D)
        _sum_into(ctx, lsegs, rsegs, tv, 0)
        emit_ln_dense(ctx, tv, D, pre + "._layernorm", tgt, L.ACT_NONE, mask)
    else:
        _sum_into(ctx, _clip_segs(lsegs, mask), _clip_segs(rsegs, mask), tgt.view, tgt.accumulate)


def _sum_into(ctx, lsegs, rsegs, dst: DV, accumulate):
    if not accumulate:
        zero_fill(ctx, dst)
    copy_into(ctx, lsegs, dst, 1)
    copy_into(ctx, rsegs, dst, 1)
    # make the zero_fill/copies visible as a producer of dst for liveness: nothing else to do


def emit_ln_dense(ctx, x: DV, D, lnname, tgt: Target, act, mask):
    """standalone LayerNorm of a dense buffer (operators that skipped their projection)"""
    def xgrad(dx):
        # dx holds dL/dx for the whole [B, D] buffer: hand it to the producers of x through x.grad
        if ctx.shape_only:
            return
        gp, acc = ctx.gtarget(x)
        d = L.CopySegsDesc()
        d.kind = L.OP_COPY_SEGS
        d.B, d.nseg, d.ld_dst, d.accumulate, d.reverse = ctx.B, 1, x.ld, acc, 0
        d.dst = gp
        d.seg[0], d.width[0], d.ld[0], d.off[0] = dx.data_ptr(), D, D, 0
        ctx.emit(d)

    emit_layernorm(ctx, L.AM_KC, x.ptr, x.ld, ctx.B, D, lnname, tgt.view.ptr, tgt.view.ld, act, mask, tgt.accumulate, tgt.view, xgrad)


def op_sigmoid_gating(ctx, cfg, pre, lsegs, Ltot, rsegs, Rtot, max_dims, dims, tgt: Target):
    """SigmoidGating modules.py:521-595: out = LN?(Linear(sigmoid(Linear_DxD(left)) * right)) * mask"""
    B = ctx.B
    D = _pad_width(lsegs, Ltot, rsegs, Rtot)
    use_ln = cfg.use_layernorm
    mask = -1 if cfg.fixed else dims
    sl = pre + "._left_self_linear._linear"
    need_proj = D != max_dims
    # rows of the [B, D] temporaries start on 128-byte boundaries (D = 13 + 1024 i: unaligned rows made the gated product write
    # 2.5x its output bytes — every 128-byte row segment a wave stores straddled two 64-byte granules)
    masked_plain = (not need_proj) and (not use_ln) and 0 <= mask < D  # the mask is all that follows the gating product
    own = need_proj or use_ln or masked_plain  # the product lands in a temporary of this operator (else straight in the target)
    Dp = (D + 31) // 32 * 32 if (own and B > 256) else D  # (the saved sigmoid is addressed like the product: same row stride)
    gbuf = ctx.buf(B * Dp, need_grad=False)
    g = DV(gbuf, 0, D, Dp, padded=True)
    if own:
        pbuf = ctx.buf(B * Dp)
        prod = DV(pbuf, 0, D, Dp, padded=True)
        ptgt = Target(prod, 0)
    else:
        prod = tgt.view
        ptgt = tgt
    dzbuf = {}

    def pre_dz():
        if not ctx.live(prod):
            return None
        dz = ctx.alloc(B * Dp)
        e = L.GateBwdDesc()
        e.kind = L.OP_GATE_BWD
        e.B, e.D = B, D
        e.ld_dout, e.ld_g, e.ld_dz = prod.ld, Dp, Dp
        e.dout, e.g, e.dz = prod.gptr, g.ptr, dz.data_ptr()
        n = 0
        for s in _live_segs(rsegs):
            gp, acc = ctx.gtarget(s.view)
            e.r_ptr[n], e.dr_ptr[n] = s.view.ptr, gp
            e.r_off[n], e.r_width[n], e.r_ld[n], e.dr_accumulate[n] = s.koff, s.width, s.view.ld, acc
            n += 1
        e.nseg = n
        ctx.emit(e)
        return dz.data_ptr(), Dp

    # the D x D self-linear consumes the zero-padded left operand: K range = Ltot (padding contributes nothing)
    linear_dense(ctx, lsegs, D, sl, D, True, prod, L.ACT_SIGMOID, -1, ptgt.accumulate, mul=rsegs, save_act=g, pre_dz_cb=pre_dz)
    if need_proj:
        linear_dense(ctx, [Seg(prod, 0, D)], D, pre + "._linear_proj", max_dims, True, tgt.view, L.ACT_NONE, mask, tgt.accumulate,
                     ln=(pre + "._layernorm") if use_ln else None)
    elif use_ln:
        emit_ln_dense(ctx, prod, D, pre + "._layernorm", tgt, L.ACT_NONE, mask)
    elif masked_plain and not ctx.shape_only:
        masked_copy(ctx, prod, mask, tgt)


def op_transformer(ctx, cfg, pre, ssegs, Ntot, max_dims, dims, out: SV):
    """Transformer modules.py:599-688"""
    B = ctx.B
    use_ln = cfg.use_layernorm
    mask = -1 if cfg.fixed else dims
    xbuf = ctx.buf(B * max_dims * E)
    x = SV(xbuf, 0, max_dims, max_dims * E)
    linear_tokens(ctx, ssegs, Ntot, pre + "._linear_proj", max_dims, not use_ln, x, L.ACT_NONE, mask,
                  ln=(pre + "._proj_ln") if use_ln else None)
    pp = [ctx.param(pre + "." + leaf, shp) for leaf, shp in zip(MHA_LEAVES, MHA_SHAPES)]
    if ctx.shape_only:
        return
    saved = ctx.alloc(B * max_dims * L.MHA_SAVED) if ctx.train else None
    d = L.MhaDesc()
    d.kind = L.OP_MHA_FWD
    d.B, d.N, d.ldx, d.ldo, d.dims_in_use = B, max_dims, x.ld, out.ld, mask
    d.x, d.out = x.ptr, out.ptr
    d.saved = saved.data_ptr() if saved is not None else None
    for q in range(12):
        d.params[q] = pp[q]
    ctx.emit(d)

    def bwd():
        if not ctx.live(out):
            return
        e = L.MhaDesc()
        e.kind = L.OP_MHA_BWD
        e.B, e.N, e.ldx, e.ldo, e.dims_in_use = B, max_dims, x.ld, out.ld, mask
        e.x, e.dout = x.ptr, out.gptr
        e.dx = xbuf.grad_tensor().data_ptr()
        xbuf.mark()
        e.saved = saved.data_ptr() if saved is not None else None
        e.bwd_form = ctx.mha_bwd_form
        for q in range(12):
            e.params[q] = pp[q]
        ctx.emit(e)
        # the per-sample parameter-gradient partials of ALL Transformer nodes share one buffer and one fixed-order
        # reduction launch at the end of the backward program (their sums feed only the optimizer): _flush_mha_reduce
        ctx.mha_reduce.append((e, [(ctx.gparam(pre + "." + leaf), MHA_OFFS[q], int(torch.Size(MHA_SHAPES[q]).numel()))
                                   for q, leaf in enumerate(MHA_LEAVES)]))

    ctx.on_backward(bwd)


def op_fm(ctx, cfg, pre, x: SV, fm_dims, dims, dense_out: DV, pre_add: Optional[DV] = None):
    """FactorizationMachine3D modules.py:720-750, added into dense_out (supernet.py:1154-1157).  With pre_add the
    projection writes dense_out = pre_add + FM term in one pass (dense_out holds nothing beforehand)."""
    B = ctx.B
    use_ln = cfg.use_layernorm
    mask = -1 if cfg.fixed else dims
    masked_direct = fm_dims == E and 0 <= mask < E  # identity projection under a prefix mask (modules.py:739-749)
    direct = fm_dims == E and not masked_direct
    if masked_direct:
        assert pre_add is None
    if direct:
        ix = dense_out
        if use_ln:  # the LayerNorm module stays registered but unused (modules.py:743)
            ctx.param(pre + "._linear_layernorm.weight", (fm_dims,), used=False)
            ctx.param(pre + "._linear_layernorm.bias", (fm_dims,), used=False)
    else:
        ixbuf = ctx.buf(B * E)
        ix = DV(ixbuf, 0, E, E)
    if not ctx.shape_only:
        d = L.FmDesc()
        d.kind = L.OP_FM_FWD
        d.B, d.N, d.ldx, d.ld_ix, d.accumulate = B, x.N, x.ld, ix.ld, 1 if (direct and pre_add is None) else 0
        d.x, d.ix = x.ptr, ix.ptr
        if direct and pre_add is not None:  # block_out = node_sum + FM term in one pass (dense_out holds nothing beforehand)
            assert pre_add.ld == ix.ld and pre_add.width == E
            d.add = pre_add.ptr
        ctx.emit(d)

        def bwd():
            if not ctx.live(ix):
                return
            gp, acc = ctx.gtarget(x)
            if gp is None:
                return
            e = L.FmDesc()
            e.kind = L.OP_FM_BWD
            e.B, e.N, e.ldx, e.ld_ix, e.accumulate = B, x.N, x.ld, ix.ld, acc
            e.x, e.dix, e.dx = x.ptr, ix.gptr, gp
            ctx.emit(e)

        ctx.on_backward(bwd)
    if masked_direct:
        if use_ln:
            ctx.param(pre + "._linear_layernorm.weight", (fm_dims,), used=False)
            ctx.param(pre + "._linear_layernorm.bias", (fm_dims,), used=False)
        if not ctx.shape_only:
            masked_copy(ctx, ix, mask, Target(dense_out, 1))
    elif not direct:
        linear_dense(ctx, [Seg(ix, 0, E)], E, pre + "._linear_proj", fm_dims, not use_ln, dense_out, L.ACT_NONE, mask,
                     0 if pre_add is not None else 1, ln=(pre + "._linear_layernorm") if use_ln else None, pre_add=pre_add)


# ----------------------------------------------------------------------------------------------------------------
# block and network walks
# ----------------------------------------------------------------------------------------------------------------
def _as_set(v):
    return set(int(x) for x in (v.tolist() if hasattr(v, "tolist") else v))


def _mk_segs(views, widths, selected, fixed):
    """ascending-j concatenation (supernet.py:536-568 / 625-633) -> (segments, total width)"""
    segs, off = [], 0
    for j, (v, w) in enumerate(zip(views, widths)):
        if j in selected:
            segs.append(Seg(v, off, w))
            off += w
        elif not fixed:
            segs.append(Seg(None, off, w))
            off += w
    return segs, off


def block_walk(ctx, cfg, pre, ops, choice, d_in, Dtot, s_in, Ntot, l_in, Ltot, r_in, Rtot):
    """SuperNetBlock.forward supernet.py:1067-1162 / fixed_forward :1185-1242.  `pre` = parameter-name prefix of the block
    ("_blocks.<i>" inside a SuperNet).  Returns (dense DV, sparse SV)."""
    B = ctx.B
    fixed = cfg.fixed
    dd, sdm = int(choice["dense_in_dims"]), int(choice["sparse_in_dims"])
    max_dense = dd if fixed else int(max(ops["dense_node_dims"]))
    max_sparse = sdm if fixed else int(max(ops["sparse_node_dims"]))
    active = [int(a) for a in choice["active_nodes"]]
    dsi = int(choice["dense_sparse_interact"])
    deep_fm = int(choice["deep_fm"])
    names = ops["node_names"]
    dense_nodes = [n for n in range(ops["num_nodes"]) if n in active and names[n] in DENSE_UNARY + DENSE_BINARY + DENSE_SPARSE]
    sparse_nodes = [n for n in range(ops["num_nodes"]) if n in active and names[n] in SPARSE_NODES]
    if not fixed:
        assert dense_nodes and sparse_nodes or True
    extra = DS_INTERACT_NUM_SPLITS if (dsi == 1 or not fixed) else 0
    sbuf = ctx.buf(B * (max_sparse + extra) * E)
    sparse_all = SV(sbuf, 0, max_sparse + extra, (max_sparse + extra) * E)
    sparse_nodes_out = sparse_all.rows(0, max_sparse)
    # dense -> sparse merge without projection (the dense output is already 8 x 16 wide): the reference's 8 extra token rows
    # are a VIEW of dense_out (no clone, supernet.py:1145-1146 / :1226-1227), and the DeepFM term is added to dense_out IN
    # PLACE afterwards (:1157 / :1236), so the rows hold the post-FM value.  Here the dense output simply LIVES in those rows
    # (a strided [B,128] view of the sparse slab): same aliasing, and neither a copy nor a gradient fan-in exists
    share_rows = bool(extra and dsi == 1 and max_dense == E * DS_INTERACT_NUM_SPLITS and not ctx.shape_only)
    if share_rows:
        dense_out = sparse_all.rows(max_sparse, DS_INTERACT_NUM_SPLITS).dense()
    else:
        dbuf = ctx.buf(B * max_dense)
        dense_out = DV(dbuf, 0, max_dense, max_dense)
    act = L.ACT_BY_NAME[cfg.activation]

    # ---- dense nodes: sum of node outputs (supernet.py:1133 / 1215) --------------------------------------------
    real_dense = [n for n in dense_nodes if names[n] != "zeros-2d"]
    n_contrib = len(real_dense) + (1 if deep_fm else 0)
    # One-pass DeepFM merge (fixed sub-networks without LayerNorm): the FM projection can write
    # block_out = pre_fm + FM term itself (GEMM pre_add), where pre_fm is the buffer holding the node sum.  That removes
    # (a) the copy of an activated node output into dense_out and (b) the copy that keeps the pre-FM value alive for the
    # dense->sparse projection; the gradient of pre_fm aliases the gradient of the block output (registered below).
    fm_dims_fixed = max_dense
    one_pass_fm = bool(deep_fm == 1 and fixed and not cfg.use_layernorm and not ctx.shape_only and not share_rows)  # (fm_dims == E: the FM kernel adds itself)
    pre_fm = None  # the buffer that holds the node sum when it is not dense_out
    wrote = False
    for n in range(ops["num_nodes"]):
        if n not in dense_nodes:
            continue
        name = names[n]
        npre = "%s._nodes.%d" % (pre, n)
        if name == "zeros-2d":
            continue
        has_act = name == "linear-2d" and act != L.ACT_NONE and not cfg.use_layernorm
        if has_act and n_contrib > 1:
            # activated output needs its own buffer for the ReLU/SiLU backward; its gradient is dense_out's gradient
            tb = ctx.buf(B * max_dense)
            tb_view = DV(tb, 0, max_dense, max_dense)
            tgt = Target(tb_view, 0)
        else:
            tgt = Target(dense_out, 1 if wrote else 0)
        if name == "linear-2d":
            op_elastic_linear(ctx, cfg, npre, d_in, Dtot, max_dense, dd, tgt)
        elif name == "dot-product":
            op_dot_product(ctx, cfg, npre, d_in, Dtot, s_in, Ntot, max_dense, dd, tgt)
        elif name == "sum":
            op_sum(ctx, cfg, npre, l_in, Ltot, r_in, Rtot, max_dense, dd, tgt)
        elif name == "sigmoid-gating":
            op_sigmoid_gating(ctx, cfg, npre, l_in, Ltot, r_in, Rtot, max_dense, dd, tgt)
        else:
            raise NotImplementedError(name)
        if tgt.view is not dense_out:
            if one_pass_fm and len(real_dense) == 1:
                pre_fm = tgt.view  # the only node: its buffer IS the node sum, dense_out is written by the FM projection
            else:
                _alias_add(ctx, tgt.view, dense_out, 1 if wrote else 0)
        wrote = True
    if not wrote:
        zero_fill(ctx, dense_out)
    node_sum = pre_fm if pre_fm is not None else dense_out  # pre-FM value of the block's dense output

    # ---- sparse nodes ------------------------------------------------------------------------------------------
    real_sparse = [n for n in sparse_nodes if names[n] != "zeros-3d"]
    wrote_s = False
    for n in range(ops["num_nodes"]):
        if n not in real_sparse:
            continue
        name = names[n]
        npre = "%s._nodes.%d" % (pre, n)
        if len(real_sparse) > 1:
            tb = ctx.buf(B * max_sparse * E)
            tv = SV(tb, 0, max_sparse, max_sparse * E)
        else:
            tv = sparse_nodes_out
        if name == "transformer":
            op_transformer(ctx, cfg, npre, s_in, Ntot, max_sparse, sdm, tv)
        elif name == "linear-3d":
            op_elastic_linear3d(ctx, cfg, npre, s_in, Ntot, max_sparse, sdm, tv)
        else:
            raise NotImplementedError(name)
        if tv is not sparse_nodes_out:
            _alias_add(ctx, tv.dense(), sparse_nodes_out.dense(), 1 if wrote_s else 0)
        wrote_s = True
    if not wrote_s:
        zero_fill(ctx, sparse_nodes_out.dense())

    # ---- dense -> sparse merge (supernet.py:1137-1150 / 1218-1231) --------------------------------------------
    if extra:
        proj_rows = sparse_all.rows(max_sparse, DS_INTERACT_NUM_SPLITS).dense()  # [B,128] view into the sparse slab
        if dsi == 1:
            if max_dense != E * DS_INTERACT_NUM_SPLITS:
                linear_dense(ctx, [Seg(node_sum, 0, max_dense)], max_dense, pre + ".project_emb_dim", E * DS_INTERACT_NUM_SPLITS,
                             not cfg.use_layernorm, proj_rows, L.ACT_NONE, -1, 0,
                             ln=(pre + ".project_emb_dim_layernorm") if cfg.use_layernorm else None)
            elif not share_rows:
                copy_into(ctx, [Seg(node_sum, 0, max_dense)], proj_rows)
        else:
            zero_fill(ctx, proj_rows)

    # ---- sparse -> dense merge: DeepFM on the node outputs only (supernet.py:1154-1157 / 1233-1236) -----------
    if deep_fm == 1:
        fm_dims = max_dense if fixed else int(max(ops["dense_node_dims"]))
        keep_pre = dsi == 1 and max_dense != E * DS_INTERACT_NUM_SPLITS  # the projection above needs the pre-FM value again (dW)
        if one_pass_fm and (pre_fm is not None or keep_pre):
            src = node_sum
            if pre_fm is None:  # node sum sits in dense_out: the block output moves to a fresh buffer
                fbuf = ctx.buf(B * max_dense)
                dense_out = DV(fbuf, 0, max_dense, max_dense)
            out_view = dense_out
            op_fm(ctx, cfg, pre + ".deep_fm", sparse_nodes_out, fm_dims, dd, out_view, pre_add=src)

            def alias_bwd():  # registered last => runs first in backward: d out / d node_sum = I, share the storage
                if not ctx.live(out_view):
                    return
                src.buf.g = out_view.buf.grad_tensor()
                src.buf.mark()

            ctx.on_backward(alias_bwd)
            return dense_out, sparse_all
        if keep_pre:
            # the projection above read dense_out *before* the FM term is added (dense_t_2d_out.clone(), supernet.py:1139)
            # and its weight gradient needs that value again: keep it, add the FM term into a copy
            fbuf = ctx.buf(B * max_dense)
            dense_final = DV(fbuf, 0, max_dense, max_dense)
            _alias_add(ctx, dense_out, dense_final, 0)
            dense_out = dense_final
        op_fm(ctx, cfg, pre + ".deep_fm", sparse_nodes_out, fm_dims, dd, dense_out)
    return dense_out, sparse_all


def _alias_add(ctx, src: DV, dst: DV, accumulate):
    """dst (+)= src where src is a private copy of a node output; in backward the node reads dst's gradient through
    src.grad := dst.grad (no copy: d dst / d src = I)."""
    if ctx.shape_only:
        return
    d = L.CopySegsDesc()
    d.kind = L.OP_COPY_SEGS
    d.B, d.nseg, d.ld_dst, d.accumulate, d.reverse = ctx.B, 1, dst.ld, accumulate, 0
    d.dst = dst.ptr
    d.seg[0], d.width[0], d.ld[0], d.off[0] = src.ptr, src.width, src.ld, 0
    ctx.emit(d)

    def bwd():
        if not ctx.live(dst):
            return
        # share the gradient storage: src's gradient view aliases dst's
        sb, db = src.buf, dst.buf
        if src.off == 0 and dst.off == 0 and src.ld == dst.ld and sb.numel == db.numel:
            sb.g = db.grad_tensor()
            sb.mark()
        else:
            gp, acc = ctx.gtarget(src)
            e = L.CopySegsDesc()
            e.kind = L.OP_COPY_SEGS
            e.B, e.nseg, e.ld_dst, e.reverse = ctx.B, 1, dst.ld, 1
            e.dst = dst.gptr
            e.seg[0], e.width[0], e.ld[0], e.off[0], e.seg_accumulate[0] = gp, src.width, src.ld, 0, acc
            ctx.emit(e)

    ctx.on_backward(bwd)


class Plan:
    """Result of compiling one (choice, B): buffers for inputs/outputs and the launch programs."""
    pass


def network_walk(ctx, cfg: NetConfig, choice, int_x: DV, sparse0: SV):
    """SuperNet.forward / fixed_forward wiring (supernet.py:513-668). Returns (dense outputs, sparse outputs) of the last
    `cfg.last_n_blocks_out` entries of the running lists — what the final layer reads (:592-596 / :657-661)."""
    dlist, slist = [int_x], [sparse0]
    for i in range(cfg.num_blocks):
        def mark(i=i):
            # registered BEFORE block i's operators, hence run AFTER all of their backward closures: block i is complete here
            # (with weight-gradient products in backward order, i.e. defer_dw off — the data-parallel bucket boundaries)
            if not ctx.defer_dw:
                _flush_mha_reduce(ctx)
                _flush_ln_reduce(ctx)
            ctx.block_marks.append((i, len(ctx.bwd)))
        ctx.on_backward(mark)
        mac = choice["macro"][i]
        dviews, dwidths = dlist, [v.width for v in dlist]
        sviews, swidths = slist, [v.N for v in slist]
        d_in, Dtot = _mk_segs(dviews, dwidths, _as_set(mac["dense_idx"]), cfg.fixed)
        s_in, Ntot = _mk_segs(sviews, swidths, _as_set(mac["sparse_idx"]), cfg.fixed)
        l_in, Ltot = _mk_segs(dviews, dwidths, _as_set(mac["dense_left_idx"]), cfg.fixed)
        r_in, Rtot = _mk_segs(dviews, dwidths, _as_set(mac["dense_right_idx"]), cfg.fixed)
        d_out, s_out = block_walk(ctx, cfg, "_blocks.%d" % i, cfg.block_ops(i), choice["micro"][i], d_in, Dtot, s_in, Ntot, l_in, Ltot, r_in, Rtot)
        dlist.append(d_out)
        slist.append(s_out)
    n = cfg.last_n_blocks_out
    return dlist[-n:], slist[-n:]


def final_segments(d_last: List[DV], s_last: List[SV]):
    """feats = cat(dense outputs, -1) ++ flatten(cat(sparse outputs, dim=-1)) (supernet.py:592-597 / 657-662) as final-logit segments:
    -> ([(Seg, tok_stride)], K).  One block: its sparse slab is one contiguous run of weight columns.  Several: the reference
    concatenates [B, N, 16] tensors on the LAST dim, so block j's token t meets weight columns D + t * (n * 16) + j * 16 + [0, 16) —
    a token-strided segment (nasrec_final_desc_t.tok_stride); torch.cat demands equal N there, and so does this."""
    segs, k = [], 0
    for v in d_last:
        segs.append((Seg(v, k, v.width), 0))
        k += v.width
    n = len(s_last)
    if len({v.N for v in s_last}) != 1:
        raise RuntimeError("Sizes of tensors must match except in dimension 2 (last_n_blocks_out = %d concatenates the sparse outputs of "
                           "the last blocks on their last dim, supernet.py:594-596): token counts %s" % (n, [v.N for v in s_last]))
    N = s_last[0].N
    for j, v in enumerate(s_last):
        segs.append((Seg(v.dense(), k + (j * E if n > 1 else 0), N * E), n * E if n > 1 else 0))
    return segs, k + n * N * E


def infer_param_shapes(cfg: NetConfig, choice, Fd, Fs, num_embeddings) -> Dict[str, tuple]:
    """Names and shapes of every parameter the reference would hold after its warm-up forward
    (train_utils.py:392-433): fixed mode walks the fixed choice, supernet mode walks the full path."""
    ctx = Ctx(B=2, device=None, params=None, grads=None, shape_only=True)
    for f in range(Fs):
        ctx.param("_embedding.%d.weight" % f, (int(num_embeddings[f]), E))
    d_last, s_last = network_walk(ctx, cfg, choice, DV(Buf(ctx, 2 * Fd, False), 0, Fd, Fd), SV(Buf(ctx, 2 * Fs * E, False), 0, Fs, Fs * E))
    _, K = final_segments(d_last, s_last)
    ctx.param("_final.weight", (1, K))
    ctx.param("_final.bias", (1,))
    return ctx.shapes


def full_path_choice(cfg: NetConfig):
    """supernet.py:814-824 + :1265-1276"""
    macro, micro = [], []
    for i in range(cfg.num_blocks):
        n = i + 1
        macro.append({k: list(range(n)) for k in ("dense_idx", "sparse_idx", "dense_left_idx", "dense_right_idx")})
        ops = cfg.block_ops(i)
        micro.append({"active_nodes": list(range(ops["num_nodes"])), "dense_in_dims": int(max(ops["dense_node_dims"])),
                      "sparse_in_dims": int(max(ops["sparse_node_dims"])), "dense_sparse_interact": 1, "deep_fm": 1})
    return {"macro": macro, "micro": micro}


def iter_ops(descs):
    """the operator descriptors of a program with worklist launches (nasrec_amd/schedule.py) expanded back into their operators
    (for FLOP counting, reports): a split-K GEMM appears once (at its main pass)"""
    for d in descs:
        if isinstance(d, L.WorklistDesc):
            for n in d.nodes:
                if n.part != "epi":
                    yield n.desc
        else:
            yield d
