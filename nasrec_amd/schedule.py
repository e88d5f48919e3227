"""Level scheduling of a launch program (batch <= 256: the latency regime).

At batch 256 every launch costs ~5 us whatever it does (cold L2, dependent round trips: DESIGN.md §3), and a step is a chain of
~60 of them.  But the step is not a chain: inside a choice block the dense nodes and the sparse nodes are independent
(supernet.py:1113-1134), a block often depends on a few of the earlier blocks only (macro choice), the backward's weight-gradient
products have no consumer before the optimizer, split-K second passes are independent of whatever runs beside them.  This module

  * derives each descriptor's read / write footprint from its fields (strided views: a [B, w] window of a row-strided slab),
  * builds the dependency DAG (RAW, WAR, WAW on overlapping footprints; program order breaks ties),
  * assigns ASAP levels, and
  * packs each level into ONE heterogeneous launch (NASREC_OP_WORKLIST, csrc/worklist.hip) whose workgroup ranges run the level's
    operators side by side; operators the worklist kernel has no body for stay launches of their own inside their level.

A split-K GEMM is cut in two schedulable pieces: the main pass (slabs out) and the second pass (fixed-order sum + epilogue), which
lands one level later beside unrelated work instead of costing a launch of its own on the critical path.

Results are bit-identical to the unscheduled program: the same bodies run on the same operands, only launch boundaries move."""
import ctypes as C
import os
from typing import List, Tuple

from . import _lib as L

E = 16
MHA_PARAM_FLOATS = [768, 48, 256, 16, 16, 16, 256, 16, 256, 16, 16, 16]


class Acc:
    """footprint: `rows` windows of `width` bytes, `stride` bytes apart, starting at ptr"""
    __slots__ = ("ptr", "width", "stride", "rows", "end")

    def __init__(self, ptr, rows, width_f, ld_f):
        self.ptr = int(ptr)
        self.rows = max(int(rows), 1)
        self.width = int(width_f) * 4
        self.stride = int(ld_f) * 4 if self.rows > 1 else self.width
        if self.stride < self.width:
            self.stride = self.width
        self.end = self.ptr + (self.rows - 1) * self.stride + self.width


def _flat(ptr, n_floats):
    return Acc(ptr, 1, n_floats, n_floats)


def overlap(a: Acc, b: Acc) -> bool:
    if a.width <= 0 or b.width <= 0 or a.end <= b.ptr or b.end <= a.ptr:
        return False
    if a.stride == b.stride and a.rows > 1 and b.rows > 1:
        s = a.stride
        d = (b.ptr - a.ptr) % s  # b's window starts d bytes into a's period
        return d < a.width or d + b.width > s
    return True  # extents overlap and the strides differ: assume the worst


def _operand(ptr, mode, R, K, ld):
    """footprint of a GEMM operand P(r, k), r < R, k < K (include/nasrec_hip.h addressing modes)"""
    if not ptr:
        return None
    if mode == L.AM_KC:
        return Acc(ptr, R, K, ld)
    if mode == L.AM_RC:
        return Acc(ptr, K, R, ld)
    if mode == L.AM_TOKR:
        return Acc(ptr, (R + 15) // 16, K * E, ld)
    return Acc(ptr, (K + 15) // 16, R * E, ld)  # TOKK


def _cview(ptr, cmode, M, N, ldc):
    if not ptr:
        return None
    if cmode == L.CM_PLAIN:
        return Acc(ptr, M, N, ldc)
    return Acc(ptr, (N + 15) // 16, M * E, ldc)


def _gemm_io(d, part):
    """part: 'whole' | 'main' (operands -> split-K slabs) | 'epi' (slabs -> epilogue -> C)"""
    R, W = [], []
    nprob = d.nseg if d.zmode else 1
    S = d.splitk if d.splitk > 1 else 1
    Mm = max(d.seg[q].M for q in range(nprob))
    Nm = max(d.seg[q].N for q in range(nprob))
    ws = _flat(d.workspace, S * Mm * Nm * nprob) if (S > 1 and d.workspace) else None
    if d.splitk == L.SPLITK_BALANCED and d.workspace:
        ws = _flat(d.workspace, L.SK_WORKSPACE_FLOATS)  # the engine-wide partial-tile workspace of the balanced schedule: shared by launches
    if part in ("whole", "main"):
        for q in range(d.nseg):
            s = d.seg[q]
            if not s.A:
                continue
            M, N = (s.M, s.N) if d.zmode else (d.seg[0].M, d.seg[0].N)
            Nb = N - (1 if s.ones_col else 0)
            for ptr, mode, rr, ld in ((s.A, d.amode, M, s.lda), (s.Aaux, d.amode, M, s.lda), (s.B, d.bmode, Nb, s.ldb), (s.Baux, d.bmode, Nb, s.ldb)):
                a = _operand(ptr, mode, rr, s.K, ld)
                if a is not None:
                    R.append(a)
        if part == "main":
            W.append(ws)
            return R, W
        if part == "whole" and ws is not None:  # a split-K product that stays one launch writes its slabs and reads them back
            W.append(ws)
            R.append(ws)
            if d.counters:
                W.append(_flat(d.counters, Mm * Nm))  # (upper bound of the arrival counters' extent)
    if part == "epi":
        R.append(ws)
    for q in range(nprob):
        s = d.seg[q]
        N = s.N - (1 if s.ones_col else 0)
        c = _cview(s.C if d.zmode else d.seg[0].C, d.cmode, s.M, N, s.ldc)
        if c is not None:
            W.append(c)
            if (s.accumulate if d.zmode else d.beta):
                R.append(c)
        if s.ones_col:
            rs = s.rowsum or d.rowsum_out
            if rs:
                W.append(_flat(rs, s.M))
    s0 = d.seg[0]
    if d.bias:
        R.append(_flat(d.bias, max(Mm, Nm)))
    for ptr, tgt in ((d.save_z, W), (d.save_act, W), (d.pre_add, R)):
        if ptr:
            tgt.append(_cview(ptr, d.cmode, s0.M, s0.N, s0.ldc))
    for q in range(d.mul_nseg):
        if d.mul_ptr[q]:
            R.append(Acc(d.mul_ptr[q], s0.M, d.mul_width[q], d.mul_ld[q]))
    return R, W


def desc_io(d, part="whole") -> Tuple[List[Acc], List[Acc]]:
    """(reads, writes) of a descriptor, or None if the kind is not modelled (the scheduler then treats it as a full barrier)"""
    k = d.kind
    R, W = [], []
    if isinstance(d, L.GemmDesc):
        R, W = _gemm_io(d, part)
    elif k in (L.OP_DOT_TRI_FWD, L.OP_DOT_TRI_BWD):
        P = d.k1 * (d.k1 - 1) // 2
        R.append(_flat(d.T, d.B * d.k1 * E))
        if k == L.OP_DOT_TRI_FWD:
            W.append(Acc(d.out, d.B, P, d.ld_out))
        else:
            R.append(Acc(d.dout, d.B, P, d.ld_out))
            W.append(_flat(d.dT, d.B * d.k1 * E))
    elif k in (L.OP_FM_FWD, L.OP_FM_BWD):
        R.append(Acc(d.x, d.B, d.N * E, d.ldx))
        if k == L.OP_FM_FWD:
            ix = Acc(d.ix, d.B, E, d.ld_ix)
            W.append(ix)
            if d.accumulate:
                R.append(ix)
            if d.add:
                R.append(Acc(d.add, d.B, E, d.ld_ix))
        else:
            R.append(Acc(d.dix, d.B, E, d.ld_ix))
            dx = Acc(d.dx, d.B, d.N * E, d.ldx)
            W.append(dx)
            if d.accumulate:
                R.append(dx)
    elif k in (L.OP_MHA_FWD, L.OP_MHA_BWD):
        R.append(Acc(d.x, d.B, d.N * E, d.ldx))
        for q in range(12):
            R.append(_flat(d.params[q], MHA_PARAM_FLOATS[q]))
        sv = _flat(d.saved, d.B * d.N * L.MHA_SAVED) if d.saved else None
        if k == L.OP_MHA_FWD:
            W.append(Acc(d.out, d.B, d.N * E, d.ldo))
            if sv:
                W.append(sv)
        else:
            R.append(Acc(d.dout, d.B, d.N * E, d.ldo))
            if sv:
                R.append(sv)
            W.append(Acc(d.dx, d.B, d.N * E, d.ldx))
            W.append(Acc(d.dparams_partial, d.B, L.MHA_PARAMS, d.partial_ld if d.partial_ld > 0 else L.MHA_PARAMS))
    elif k == L.OP_REDUCE_ROWS:
        R.append(Acc(d.in_, d.R, d.C, d.ld))
        for q in range(d.ndst):
            if d.dst[q]:
                W.append(_flat(d.dst[q], d.dst_len[q]))
    elif k == L.OP_COPY_SEGS:
        for q in range(d.nseg):
            win = Acc(d.dst + 4 * d.off[q], d.B, d.width[q], d.ld_dst)
            seg = Acc(d.seg[q], d.B, d.width[q], d.ld[q]) if d.seg[q] else None
            if not d.reverse:
                W.append(win)
                if d.accumulate:
                    R.append(win)
                if seg:
                    R.append(seg)
            elif seg:
                R.append(win)
                W.append(seg)
                if d.seg_accumulate[q]:
                    R.append(seg)
    elif k == L.OP_GATE_BWD:
        R += [Acc(d.dout, d.B, d.D, d.ld_dout), Acc(d.g, d.B, d.D, d.ld_g)]
        W.append(Acc(d.dz, d.B, d.D, d.ld_dz))
        for q in range(d.nseg):
            if d.r_ptr[q]:
                R.append(Acc(d.r_ptr[q], d.B, d.r_width[q], d.r_ld[q]))
            if d.dr_ptr[q]:
                a = Acc(d.dr_ptr[q], d.B, d.r_width[q], d.r_ld[q])
                W.append(a)
                if d.dr_accumulate[q]:
                    R.append(a)
    elif k == L.OP_ROWSUM:
        R.append(_operand(d.p, d.mode, d.R, d.K, d.ld))
        if d.aux:
            R.append(_operand(d.aux, d.mode, d.R, d.K, d.ld))
        W.append(_flat(d.out, d.R))
    elif k in (L.OP_FINAL_FWD, L.OP_FINAL_BWD, L.OP_FINAL_FUSED):
        def extent(q):  # weight columns a segment spans (token-strided for last_n_blocks_out > 1: nasrec_final_desc_t.tok_stride)
            ts, Wq = d.tok_stride[q], d.width[q]
            return ((Wq - 1) // 16) * ts + (Wq - 1) % 16 + 1 if (ts and Wq > 0) else Wq
        K = max([d.off[q] + extent(q) for q in range(d.nseg)] + [0])
        for q in range(d.nseg):
            if d.seg[q]:
                R.append(Acc(d.seg[q], d.B, d.width[q], d.ld[q]))
        R.append(_flat(d.w, K))
        if k in (L.OP_FINAL_FWD, L.OP_FINAL_FUSED):
            R.append(_flat(d.bias, 1))
            W.append(_flat(d.logits, d.B))
        if k == L.OP_FINAL_FUSED:
            R.append(_flat(d.y, d.B))
        if k == L.OP_FINAL_BWD:
            for ptr in (d.dlogits, d.logits, d.y):
                if ptr:
                    R.append(_flat(ptr, d.B))
        if k == L.OP_FINAL_FUSED or (k == L.OP_FINAL_BWD and not d.dseg_done):
            for q in range(d.nseg):
                if d.dseg[q]:
                    a = Acc(d.dseg[q], d.B, d.width[q], d.ld[q])
                    W.append(a)
                    if d.dseg_accumulate[q]:
                        R.append(a)
        if k == L.OP_FINAL_BWD:
            W.append(_flat(d.dw, (K + 1) * max(d.nsplit, 1)))
            for ptr, n in ((d.dbias, 1), (d.loss, 1), (d.dlogits_out, d.B)):
                if ptr:
                    W.append(_flat(ptr, n))
    elif k == L.OP_BCE:
        R += [_flat(d.logits, d.B), _flat(d.y, d.B)]
        W += [_flat(d.loss, 1), _flat(d.dlogits, d.B)]
    elif k in (L.OP_LAYERNORM_FWD, L.OP_LAYERNORM_BWD):
        if d.mode == L.AM_KC:
            vx = lambda p, ld: Acc(p, d.R, d.D, ld)
            ndx = d.R * d.ldx
        else:
            vx = lambda p, ld: Acc(p, d.R // 16, d.D * E, ld)
            ndx = (d.R // 16) * d.ldx
        R += [vx(d.x, d.ldx), _flat(d.w, d.D), _flat(d.b, d.D)]
        st = _flat(d.stats, d.R * 2)
        if k == L.OP_LAYERNORM_FWD:
            y = vx(d.y, d.ldy)
            W += [y, st]
            if d.accumulate:
                R.append(y)
        else:
            R += [st, vx(d.dy, d.ldy)]
            W += [_flat(d.dx, ndx), _flat(d.dwb_partial, d.nblk * 2 * d.D)]
    elif k == L.OP_ACT_BWD:
        if d.mode == L.AM_KC:
            v = lambda p, ld: Acc(p, d.R, d.D, ld)
        else:
            v = lambda p, ld: Acc(p, d.R // 16, d.D * E, ld)
        R += [v(d.dy, d.ld_dy), v(d.z, d.ld_z)]
        W.append(v(d.dz, d.ld_dz))
    elif k == L.OP_DEDUP_IDS:  # ids in; leaders / runs / lists out (the optimizer's launches behind the program read them)
        R.append(_flat(d.idx, d.B * d.Fs * 2))
        W += [_flat(d.leader, d.B * d.Fs), _flat(d.order, d.Fs * d.cap), _flat(d.lists, d.Fs * d.cap), _flat(d.counts, d.Fs * 2)]
        if d.heads:
            W.append(_flat(d.heads, d.Fs * d.cap))
    elif k == L.OP_MEMSET and not d.chunks:
        W.append(_flat(d.ptr, (d.bytes + 3) // 4))
    else:
        return None
    return [a for a in R if a is not None and a.ptr], [a for a in W if a is not None and a.ptr]


class Node:
    __slots__ = ("desc", "part", "reads", "writes", "level", "index")

    def __init__(self, desc, part="whole"):
        self.desc, self.part = desc, part
        io = desc_io(desc, part)
        self.reads, self.writes = io if io is not None else (None, None)
        self.level = 0
        self.index = 0


def expand(descs) -> List[Node]:
    """program -> schedulable nodes: split-K GEMMs become (main pass, second pass)"""
    nodes = []
    for d in descs:
        if isinstance(d, L.GemmDesc) and d.splitk > 1 and not d.defer_second_pass:
            nodes.append(Node(d, "main"))
            nodes.append(Node(d, "epi"))
        else:
            nodes.append(Node(d))
    for i, n in enumerate(nodes):
        n.index = i
    return nodes


def _depends(later: Node, earlier: Node) -> bool:
    if later.reads is None or earlier.reads is None:
        return True  # an unmodelled kind orders everything around it
    for w in earlier.writes:
        for a in later.reads:
            if overlap(w, a):
                return True
        for a in later.writes:
            if overlap(w, a):
                return True
    for r in earlier.reads:
        for a in later.writes:
            if overlap(r, a):
                return True
    return False


def assign_levels(nodes: List[Node]) -> int:
    """ASAP level of every node (program order respected between dependent nodes); returns the number of levels"""
    for i, n in enumerate(nodes):
        lv = 0
        for j in range(i - 1, -1, -1):
            m = nodes[j]
            if m.level + 1 > lv and _depends(n, m):
                lv = m.level + 1
        n.level = lv
    return (max(n.level for n in nodes) + 1) if nodes else 0


# ----------------------------------------------------------------------------------------------------------------
# slack: ASAP levels put every operator as early as its operands allow, but a level lasts as long as its slowest member plus
# whatever the chip cannot hold at once — and many operators are in no hurry: weight-gradient products and their second passes
# (nobody reads a weight gradient before the optimizer), bias / parameter reductions, forward operators whose result no later block
# selects.  `balance_levels` moves such nodes, inside the window their dependencies leave, to the level where they cost least:
# typically beside a Transformer backward (one wavefront per SIMD for 17-20 us: the matrix pipes are idle) instead of beside the
# two large products of the step's critical path.  Same bodies on the same operands: bit-identical results.
# ----------------------------------------------------------------------------------------------------------------
BALANCE = os.environ.get("NASREC_WL_BALANCE", "1") != "0"
_PUSH = os.environ.get("NASREC_WL_PUSH", "1") != "0"  # (A/B knob: 0 = a node only moves inside the window its successors leave as they stand)
_LATENCY_NS = int(os.environ.get("NASREC_WL_LAT_NS", "5000"))  # what a level costs however little it does (launch, descriptor + operand first touch on cold caches, drain)
_GEMM_DIV = int(os.environ.get("NASREC_WL_GEMM_DIV", "20000"))  # M N K per ns of a product on the worklist tile (A/B knob of the balancing model)


_BAL_R4 = os.environ.get("NASREC_WL_BAL_COST", "r4") == "r4"  # (A/B knob)


def _BAL_COST(node):
    return _cost_r4(node) if _BAL_R4 else _cost(node)


def _work_ns(node):
    """the part of an item's stand-alone duration that occupies the chip (adds up when items share a level), as opposed to latency
    (which overlaps): a product's M*N*K term; a quarter of the sample-per-workgroup bodies (one wavefront per SIMD)"""
    c = _BAL_COST(node)
    if isinstance(node.desc, L.GemmDesc):
        return max(c - _LATENCY_NS, 0) if node.part != "epi" else 400
    if node.desc.kind in (L.OP_MHA_FWD, L.OP_MHA_BWD):
        return c // _MHA_NS[4 if node.desc.kind == L.OP_MHA_FWD else 5]
    return c // 4


_MHA_PAIR = os.environ.get("NASREC_WL_MHA_PAIR", "1") != "0"  # (A/B knob)


def _level_ns(members):
    """estimated duration of a level's worklist launch.  A Transformer forward beside a Transformer backward is nearly free: both are one
    wavefront per SIMD walking barrier-separated stages, two such workgroups per CU fill each other's waits — whereas in a short level
    of products the forward's own 9-11 us become the level's duration."""
    if not members:
        return 0
    pair = _MHA_PAIR and any(n.desc.kind == L.OP_MHA_BWD for n in members)
    work = sum((_BAL_COST(n) // 12 if pair and n.desc.kind == L.OP_MHA_FWD else _work_ns(n)) for n in members)
    return max(max(_BAL_COST(n) for n in members), _LATENCY_NS + work)


def balance_levels(nodes: List[Node], nl: int) -> None:
    """in place: nodes with slack move to the level of their window that makes the estimated sum of level durations smallest (local
    search from the ASAP schedule; the number of levels never grows).  A move to level lv PUSHES the node's successors that sit at
    or before lv to the level behind their predecessor, recursively, as far as their own successors allow: a Transformer forward
    (11 us) with one level of slack shares its ASAP level with a critical DotProduct core (6 us) — it can only leave if the
    token-axis products reading its output step back a level too, which they can."""
    n = len(nodes)
    pred = [[] for _ in range(n)]
    succ = [[] for _ in range(n)]
    for i in range(n):
        for j in range(i):
            if _depends(nodes[i], nodes[j]):
                pred[i].append(j)
                succ[j].append(i)
    idx = {id(nd): i for i, nd in enumerate(nodes)}
    movable = set(i for i in range(n) if nodes[i].reads is not None and item_bytes(nodes[i]) is not None)
    levels = [[] for _ in range(nl)]
    for i in movable:
        levels[nodes[i].level].append(nodes[i])
    order = sorted(movable, key=lambda i: -_BAL_COST(nodes[i]))
    size = {id(nodes[i]): (len(item_bytes(nodes[i])) + 15) & ~15 for i in movable}

    def plan_move(i, lv, moves):
        """node i to level lv, successors pushed behind it: fills moves {node index: level}; False if some node cannot go"""
        if lv > nl - 1 or (i not in movable and nodes[i].level != lv):
            return False
        moves[i] = lv
        for j in succ[i]:
            cur = moves.get(j, nodes[j].level)
            if cur <= lv:
                if not _PUSH:
                    return False
                # a Transformer forward that found its place beside a Transformer backward stays there (see _level_ns)
                if _MHA_PAIR and nodes[j].desc.kind == L.OP_MHA_FWD and any(m.desc.kind == L.OP_MHA_BWD for m in levels[nodes[j].level]):
                    return False
                if not plan_move(j, lv + 1, moves):
                    return False
        return True

    def delta(moves):
        """change of the estimated total if `moves` are applied, or None if a launch would overflow"""
        touched = {}
        for i, lv in moves.items():
            nd = nodes[i]
            if lv == nd.level:
                continue
            touched.setdefault(nd.level, [list(levels[nd.level]), None])
            touched.setdefault(lv, [list(levels[lv]), None])
        if not touched:
            return None
        for i, lv in moves.items():
            nd = nodes[i]
            if lv == nd.level:
                continue
            touched[nd.level][0] = [m for m in touched[nd.level][0] if m is not nd]
            touched[lv][0].append(nd)
        d = 0
        for lv, (mem, _) in touched.items():
            if len(mem) > L.WL_MAX_ITEMS or sum(size[id(m)] for m in mem) > L.WL_BLOB_BYTES:
                return None  # one launch per level: a second one would cost a boundary
            d += _level_ns(mem) - _level_ns(levels[lv])
        return d

    for _ in range(4):
        moved = False
        for i in order:
            nd = nodes[i]
            lo = max([nodes[j].level + 1 for j in pred[i]] + [0])
            best, best_moves = -200, None  # (only moves worth more than the model's noise)
            for lv in range(lo, nl):
                if lv == nd.level:
                    continue
                moves = {}
                if not plan_move(i, lv, moves):
                    break  # (later levels push even further)
                d = delta(moves)
                if d is not None and d < best:
                    best, best_moves = d, moves
            if best_moves:
                for k, lv in best_moves.items():
                    m = nodes[k]
                    if lv != m.level:
                        levels[m.level] = [x for x in levels[m.level] if x is not m]
                        levels[lv].append(m)
                        m.level = lv
                moved = True
        if not moved:
            break


# measured (tools/r04_ab.sh resplit, cfg 2, three runs each): 0.3013 / 0.3021 / 0.3026 ms with the re-cut against 0.3021 / 0.3025 / 0.3014
# without — the worklist's 32x32 tile runs the 0.6 GFLOP product at ~30 TFLOP/s where the one-pass kernel reaches 46, which eats what
# the overlap with a Transformer-backward level gains.  Off by default; NASREC_WL_RESPLIT=1 turns it on.
RESPLIT = os.environ.get("NASREC_WL_RESPLIT", "0") == "1"
RESPLIT_MIN_SLACK = 3


def resplit_slack_solos(descs, alloc, min_slack=None):
    """A large forward product that runs as a launch of its own (csrc/gemm_kslice.hip: K split inside the workgroup) sits alone on the
    stream for 9-13 us.  When nothing waits for its result for several levels (in ea_criteo_kaggle_xlarge_best_1shot.json: all of
    block 5, which no later block selects — computed by the reference and thrown away), it is worth more as a worklist item beside a
    latency-bound level (a Transformer backward keeps one wavefront per SIMD busy for 17-20 us): the product is re-cut with the
    ordinary split-K factor of its tile count (workspace from `alloc`), so that its main pass and second pass become schedulable
    items.  -> new descriptor list (the same objects where nothing changed).  The re-cut product sums its k-slices in another order
    than the one-pass kernel: it is only applied where slack >= RESPLIT_MIN_SLACK levels says the consumer, if any, is far away."""
    from . import plan as P
    nodes = expand_for_worklists(descs)
    nl = assign_levels(nodes)
    n = len(nodes)
    alap = [nl - 1] * n
    for i in range(n - 1, -1, -1):
        for j in range(i + 1, n):
            if alap[j] - 1 < alap[i] and _depends(nodes[j], nodes[i]):
                alap[i] = alap[j] - 1
    out, changed = list(descs), False
    for i, nd in enumerate(nodes):
        d = nd.desc
        if nd.part != "whole" or not isinstance(d, L.GemmDesc) or d.splitk > 1 or alap[i] - nd.level < (RESPLIT_MIN_SLACK if min_slack is None else min_slack):
            continue
        if P.gemm_kernel_name(d) != "gemm_kslice_kernel":
            continue
        s0 = d.seg[0]
        kt = sum((d.seg[q].K + 31) // 32 for q in range(d.nseg) if d.seg[q].A)
        S = P._splitk_for(((s0.M + 63) // 64) * ((s0.N + 63) // 64), kt)
        if S <= 1:
            continue
        d2 = L.GemmDesc.from_buffer_copy(d)
        d2.splitk = S
        ws = alloc(S * s0.M * s0.N)
        d2.workspace = ws.data_ptr()
        d2._wl_force, d2._keep = True, ws
        out[next(k for k, x in enumerate(out) if x is d)] = d2
        changed = True
    return out if changed else descs


def levels_of(descs):
    nodes = expand(descs)
    nl = assign_levels(nodes)
    out = [[] for _ in range(nl)]
    for n in nodes:
        out[n.level].append(n)
    return out


# ----------------------------------------------------------------------------------------------------------------
# packing levels into NASREC_OP_WORKLIST launches
# ----------------------------------------------------------------------------------------------------------------
_GEMM_HEAD = L.GemmDesc.seg.offset
_SEG_BYTES = C.sizeof(L.GemmSeg)
_PLAIN_KINDS = (L.OP_MHA_FWD, L.OP_MHA_BWD, L.OP_FM_FWD, L.OP_FM_BWD, L.OP_DOT_TRI_FWD, L.OP_DOT_TRI_BWD, L.OP_COPY_SEGS, L.OP_GATE_BWD,
                L.OP_FINAL_FWD, L.OP_FINAL_BWD, L.OP_DEDUP_IDS, L.OP_FINAL_FUSED)
WL_LDS_FLOATS = 1696 + 5 * 1024 + 1024  # csrc/worklist_body.h WL_LDS_FLOATS


# measured on cfg 2 (ms per step): 128 -> 0.420, 160 -> 0.415, 260 -> 0.408, never solo -> 0.409: the two large backward products of
# the dominant Linear (dx: 208 workgroups with its split-K, dW: 156) are worth more side by side in their level's worklist than each
# alone on its better kernel
SOLO_TILES = int(os.environ.get("NASREC_WL_SOLO_TILES", "256"))


def gemm_capable(d) -> bool:
    """mirror of csrc/worklist.hip wl_gemm_geometry: can this GEMM launch run as a worklist item?"""
    if d.splitk == L.SPLITK_BALANCED or d.defer_second_pass:
        return False
    kca, kcb = d.amode in (L.AM_KC, L.AM_TOKK), d.bmode in (L.AM_KC, L.AM_TOKK)
    if not kca and kcb:
        return False
    if d.splitk > 1 and not d.workspace:
        return False
    # one large product (>= SOLO_TILES workgroups of 64x64) fills the chip on its own and is faster in its own kernel than on the
    # worklist kernel's small tiles
    nprob = d.nseg if d.zmode else 1
    S = d.splitk if d.splitk > 1 else 1
    if nprob == 1 and ((d.seg[0].M + 63) // 64) * ((d.seg[0].N + 63) // 64) * S >= SOLO_TILES:
        return False
    from . import plan as P
    if not getattr(d, "_wl_force", False) and P.kslice_eligible(d.amode, d.bmode, d.cmode, [dict(A=d.seg[q].A, Aaux=d.seg[q].Aaux, Baux=d.seg[q].Baux, ones_col=d.seg[q].ones_col,
                                                           Mvalid=d.seg[q].Mvalid, M=d.seg[q].M, N=d.seg[q].N, K=d.seg[q].K, lda=d.seg[q].lda,
                                                           ldb=d.seg[q].ldb) for q in range(d.nseg)], d.zmode):
        return False  # csrc/gemm_kslice.hip: a kernel of its own (1024-thread workgroups, 133 KB of LDS)
    s0 = d.seg[0]
    if not d.zmode:
        for q in range(d.nseg):
            s = d.seg[q]
            if s.ones_col != s0.ones_col or s.Mvalid != s0.Mvalid:
                return False
            if s.A and (bool(s.Aaux) != bool(s0.Aaux) or bool(s.Baux) != bool(s0.Baux)):
                return False
    return True


def item_bytes(node):
    """the bytes a worklist item carries for this node, or None if the worklist kernel has no body for it"""
    d = node.desc
    k = d.kind
    if isinstance(d, L.GemmDesc):
        if not gemm_capable(d):
            return None
        return C.string_at(C.addressof(d), _GEMM_HEAD + d.nseg * _SEG_BYTES)
    if k in (L.OP_DOT_TRI_FWD, L.OP_DOT_TRI_BWD):
        P4 = (d.k1 * (d.k1 - 1) // 2 + 3) & ~3
        if d.k1 < 2 or 4 * (d.k1 * 20 + P4) > WL_LDS_FLOATS:
            return None
    if k in (L.OP_MHA_FWD, L.OP_MHA_BWD) and (d.N < 1 or d.N > 64 or (k == L.OP_MHA_BWD and not d.saved)):
        return None
    if k == L.OP_DEDUP_IDS and (d.B > 256 or d.cap != 256):
        return None
    if k in _PLAIN_KINDS:
        return C.string_at(C.addressof(d), C.sizeof(d))
    if k == L.OP_REDUCE_ROWS and 1 <= d.ndst <= L.WL_REDUCE_DST:
        r = L.WlReduce()
        r.kind, r.R, r.C, r.ld, r.in_, r.ndst = d.kind, d.R, d.C, d.ld, d.in_, d.ndst
        for q in range(d.ndst):
            r.dst[q], r.dst_off[q], r.dst_len[q] = d.dst[q], d.dst_off[q], d.dst_len[q]
        return C.string_at(C.addressof(r), C.sizeof(r))
    return None


_PART = {"whole": L.WL_WHOLE, "main": L.WL_MAIN, "epi": L.WL_EPI}


# stand-alone times of the non-GEMM items of the batch-256 step, ns (tools/step_table.py ITEMS=11)
_ITEM_NS = {L.OP_MHA_BWD: 18000, L.OP_MHA_FWD: 11000, L.OP_DOT_TRI_BWD: 10000, L.OP_DOT_TRI_FWD: 6000, L.OP_REDUCE_ROWS: 4500,
            L.OP_FM_BWD: 3900, L.OP_FM_FWD: 3500, L.OP_FINAL_BWD: 7500, L.OP_DEDUP_IDS: 8000, L.OP_FINAL_FUSED: 4500}
# round 4 (ITEMS=7 python tools/step_table.py on the balanced plan): the same items measured again, used by the balancing pass
_ITEM_NS_R4 = {L.OP_MHA_BWD: 18500, L.OP_MHA_FWD: 10500, L.OP_REDUCE_ROWS: 6000, L.OP_FM_BWD: 7000, L.OP_FM_FWD: 3500, L.OP_FINAL_BWD: 7000,
               L.OP_FINAL_FWD: 3400, L.OP_GATE_BWD: 3500, L.OP_DEDUP_IDS: 8000, L.OP_FINAL_FUSED: 4500}
_COST_MODEL = os.environ.get("NASREC_WL_COST", "time")
# round 6: the Transformer items' durations in the balancing model, ns: base + slope * N for the forward and the backward, and the divisor
# of a body's duration that counts as chip occupancy when it shares a level.  The token-major bodies (csrc/attention_tok.h) take 4.3 /
# 8.4 us forward and 7.2 / 13.8 us backward on their own (N = 8 / 64; tools/mha_bench.py), but WHAT the pass moves beside them depends on
# these numbers, and the step was timed over a grid of them (tools/sweep_mha_cost.sh, tools/run_env_ab.sh, three runs each): measured
# durations 0.2465 ms, round 5's constants (10.5 / 18.5 us) 0.2224, 10.5 / 16 us 0.2214, 12 / 16 0.226, 9.5 / 16 0.229 — the model's
# "latency + sum of work" form is too coarse for these numbers to mean what they say; they are tuning constants.
_MHA_NS = [int(v) for v in os.environ.get("NASREC_WL_MHA_NS", "10500,0,16000,0,4,4").split(",")]
# settings the engine's compile-time tuner tries besides the default (engine._tune_levels): (forward base, slope, backward base, slope)
TUNE_MHA_NS = [(f, 0, b, 0) for f in (8000, 10500) for b in (12000, 14000, 16000, 18500, 21000, 24000)] + [(6000, 0, 24000, 0), (3500, 95, 6400, 105)]


def _mha_ns(d):
    return _MHA_NS[0] + _MHA_NS[1] * d.N if d.kind == L.OP_MHA_FWD else _MHA_NS[2] + _MHA_NS[3] * d.N


def _cost(node):
    """rough duration of an item on its own, ns: the long items of a level go first in its launch (workgroups are dispatched in
    blockIdx order; a long item queued behind the short ones starts late and the level ends late)"""
    d = node.desc
    if _COST_MODEL == "size":  # (the round-2 order: products by size, everything else behind them)
        if isinstance(d, L.GemmDesc):
            return 1 if node.part == "epi" else sum(d.seg[q].M * d.seg[q].N * max(d.seg[q].K, 1) for q in range(d.nseg) if d.seg[q].A)
        if d.kind in (L.OP_MHA_FWD, L.OP_MHA_BWD):
            return d.B * d.N * 16 * 4000 * (2 if d.kind == L.OP_MHA_BWD else 1)
        return 1000
    if isinstance(d, L.GemmDesc):
        if node.part == "epi":
            return 3600
        return 5000 + sum(d.seg[q].M * d.seg[q].N * max(d.seg[q].K, 1) for q in range(d.nseg) if d.seg[q].A) // 20000
    if d.kind == L.OP_FINAL_BWD and d.dseg_done:
        return 5000
    if d.kind in (L.OP_MHA_FWD, L.OP_MHA_BWD):
        return _mha_ns(d)
    return _ITEM_NS.get(d.kind, 3000)


def _cost_r4(node):
    """the balancing pass's duration model (the item ORDER inside a launch keeps `_cost`: its A/B stands): products pay ~0.9 us per
    extra k-segment (a token-axis Linear over four segments 8.7 us, over one 5.7), the DotProduct cores scale with k1^2"""
    d = node.desc
    if isinstance(d, L.GemmDesc):
        if node.part == "epi":
            return 4000
        live = [q for q in range(d.nseg) if d.seg[q].A]
        extra = 0 if d.zmode else 900 * max(len(live) - 1, 0)
        return 5200 + extra + sum(d.seg[q].M * d.seg[q].N * max(d.seg[q].K, 1) for q in live) // _GEMM_DIV
    if d.kind == L.OP_DOT_TRI_BWD:
        return 3500 + int(2.3 * d.k1 * d.k1)
    if d.kind == L.OP_DOT_TRI_FWD:
        return 3000 + int(0.2 * d.k1 * d.k1)
    if d.kind == L.OP_FINAL_BWD and d.dseg_done:
        return 5000
    if d.kind in (L.OP_MHA_FWD, L.OP_MHA_BWD):
        return _mha_ns(d)
    return _ITEM_NS_R4.get(d.kind, 3000)


def expand_for_worklists(descs) -> List[Node]:
    """as expand(), but a split-K GEMM is cut in two only when both halves can ride in worklist launches"""
    nodes = []
    for d in descs:
        if isinstance(d, L.GemmDesc) and d.splitk > 1 and gemm_capable(d):
            nodes.append(Node(d, "main"))
            nodes.append(Node(d, "epi"))
        else:
            nodes.append(Node(d))
    for i, n in enumerate(nodes):
        n.index = i
    return nodes


_SOLO_ITEMS = os.environ.get("NASREC_WL_SOLO_ITEMS", "1") != "0"  # (A/B knob)


def _item_beats_kernel(d):
    """a level's only operator normally runs as its stand-alone kernel; the token-axis input gradients (W^T dy per input segment, K <= 64
    rows of the Linear, mask on dy allowed) are the exception: the worklist's wavefront-per-tile body (csrc/worklist_body.h wl_token_dx)
    takes 3.6 - 5 us where the general template takes 5.5 - 8.9 (tools/step_table.py ITEMS=1; the conditions of csrc/worklist.hip)"""
    if not (_SOLO_ITEMS and isinstance(d, L.GemmDesc) and d.zmode and d.splitk <= 1):
        return False
    if (d.amode, d.bmode, d.cmode) != (L.AM_RC, L.AM_TOKR, L.CM_TOKJ):
        return False
    n0 = d.seg[0].N
    for q in range(d.nseg):
        sg = d.seg[q]
        if sg.Aaux or sg.ones_col or (0 < sg.Mvalid < sg.M) or sg.N != n0 or sg.K > 64 or sg.M <= 0:
            return False
    return n0 % 16 == 0


def pack(descs, alloc=None):
    """program -> scheduled program: per level, the operators the worklist kernel has bodies for share NASREC_OP_WORKLIST launches
    (as many as their descriptors need blobs), the others stay launches of their own.  Returns (new descriptor list, number of
    levels).  Every WorklistDesc carries `.nodes` (its items' Nodes) for reports."""
    if BALANCE and RESPLIT and alloc is not None:
        descs = resplit_slack_solos(descs, alloc)
    nodes = expand_for_worklists(descs)
    nl = assign_levels(nodes)
    if BALANCE:
        balance_levels(nodes, nl)
    out = []
    for lv in range(nl):
        members = sorted((n for n in nodes if n.level == lv), key=lambda n: -_cost(n))
        solo, items = [], []
        for n in members:
            b = item_bytes(n)
            (items if b is not None else solo).append((n, b))
        for n, _ in solo:
            assert n.part == "whole"
            out.append(n.desc)
        if len(items) == 1 and items[0][0].part == "whole" and not _item_beats_kernel(items[0][0].desc):
            out.append(items[0][0].desc)  # nothing to share a launch with: the stand-alone kernel
            continue
        cur, off = None, 0
        for n, b in items:
            size = (len(b) + 15) & ~15
            if cur is None or cur.n == L.WL_MAX_ITEMS or off + size > L.WL_BLOB_BYTES:
                cur = L.WorklistDesc()
                cur.kind, cur.n = L.OP_WORKLIST, 0
                cur.nodes = []
                off = 0
                out.append(cur)
            it = cur.item[cur.n]
            it.kind, it.part, it.off = n.desc.kind, _PART[n.part], off
            C.memmove(C.addressof(cur) + L.WorklistDesc.blob.offset + off, b, len(b))
            cur.n += 1
            cur.nodes.append(n)
            off += size
    return out, nl


# ----------------------------------------------------------------------------------------------------------------
# the persistent form (round 6): the items of MANY levels in one NASREC_OP_PERSIST launch, dependencies resolved in the kernel
# ----------------------------------------------------------------------------------------------------------------
PERSIST = os.environ.get("NASREC_PERSIST", "1") != "0"
_PS_ORDER = os.environ.get("NASREC_PERSIST_ORDER", "start")  # level: (level, longest first); start: estimated start time with slack (below)
_PS_ALPHA = float(os.environ.get("NASREC_PERSIST_ALPHA", "0.5"))
PERSIST_RESPLIT = os.environ.get("NASREC_PERSIST_RESPLIT", "off")
_PS_THROTTLE = float(os.environ.get("NASREC_PERSIST_THROTTLE", "0.5"))  # fraction of an operator's slack window its workgroups may take (0 = no throttling)  # off | slack | all: stand-alone products re-cut into items of the persistent launch


class PersistRefused(Exception):
    """the program cannot run in the persistent form (the caller falls back to one launch per level); str(e) says why"""


def _inside(a: Acc, ranges) -> bool:
    return any(lo <= a.ptr and a.end <= hi for lo, hi in ranges)


def _visible_edge(later: Node, earlier: Node, uc_ranges):
    """what `later` must SEE of `earlier` (read-after-write) or overwrite in order (write-after-write): every such footprint of
    `earlier` has to lie in uncached memory — cached lines written by two workgroups of one launch on different XCDs are neither
    visible to each other nor written back in order.  (Write-after-read needs ordering only.)  -> the first offending footprint or None"""
    for w in earlier.writes:
        if any(overlap(w, a) for a in later.reads) or any(overlap(w, a) for a in later.writes):
            if not _inside(w, uc_ranges):
                return w
    return None


def pack_persistent(descs, dev_alloc, uc_ranges, alloc=None):
    """program -> scheduled program whose worklist-capable operators run as items of NASREC_OP_PERSIST launches: one launch per maximal
    run of levels without a stand-alone kernel in between.  dev_alloc(nbytes) -> device tensor (tables, counters, flags; kept alive on
    the descriptor); uc_ranges: [(lo, hi)] address ranges of the plan's uncached arena.  Returns (descriptor list, number of levels);
    raises PersistRefused when a dependency edge runs through cached memory or an item has too many dependencies."""
    # A product that would stand alone on the stream (csrc/gemm_kslice.hip: 1024-thread workgroups, no body in this kernel) cuts the persistent
    # launch in two — a full drain and refill of the chip on either side of it.  Re-cut into split-K items it rides inside the launch:
    # "slack" = only where nothing waits for it for several levels (the dead block's 256 x 768 x 1565 product: 13 us of the step that then
    # overlap with the backward), "all" = every such product.  (A re-cut product sums its k-slices in another order than the one-pass kernel.)
    if alloc is not None and PERSIST_RESPLIT != "off":
        descs = resplit_slack_solos(descs, alloc, 0 if PERSIST_RESPLIT == "all" else RESPLIT_MIN_SLACK)
    nodes = expand_for_worklists(descs)
    nl = assign_levels(nodes)
    if BALANCE:
        balance_levels(nodes, nl)
    n = len(nodes)
    pred = [[j for j in range(i) if _depends(nodes[i], nodes[j])] for i in range(n)]
    out = []
    seg = []  # node indices of the current segment, in launch order

    def flush():
        if not seg:
            return
        if len(seg) == 1 and nodes[seg[0]].part == "whole":
            out.append(nodes[seg[0]].desc)  # nothing to overlap with: the stand-alone kernel
            seg.clear()
            return
        pos = {i: k for k, i in enumerate(seg)}
        inseg = set(seg)
        uc = uc_ranges() if callable(uc_ranges) else uc_ranges  # (now: the re-cut products' workspaces may have grown the arena)
        items = (L.PersistItem * len(seg))()
        blob = bytearray()
        for k, i in enumerate(seg):
            nd = nodes[i]
            b = item_bytes(nd)
            it = items[k]
            it.kind, it.part, it.off = nd.desc.kind, _PART[nd.part], len(blob)
            if hint[i] < 0:
                # units of this operator: unknown here (the launcher's geometry); the window travels as "workgroups that finish in it at 5 us per unit":
                # resolved below by a dry geometry pass
                it._pad[1] = hint[i]
            blob += b
            blob += bytes((-len(blob)) % 16)
            ps = [j for j in pred[i] if j in inseg]
            # transitive reduction: a predecessor that another predecessor already waits for (directly or not) needs no edge of its own
            anc = {}
            def ancestors(j):
                if j not in anc:
                    a = set()
                    for q in pred[j]:
                        if q in inseg:
                            a.add(q)
                            a |= ancestors(q)
                    anc[j] = a
                return anc[j]
            keep = [j for j in ps if not any(j in ancestors(q) for q in ps if q != j)]
            if len(keep) > L.PS_MAX_DEPS:
                raise PersistRefused("an operator waits for %d others (limit %d)" % (len(keep), L.PS_MAX_DEPS))
            for j in ps:  # (visibility is checked on EVERY edge, reduced or not: the data of an indirect predecessor is read all the same)
                bad = _visible_edge(nd, nodes[j], uc)
                if bad is not None:
                    raise PersistRefused("kind %d reads or overwrites [%#x, %#x) written by kind %d in the same launch, outside the uncached arena"
                                         % (nd.desc.kind, bad.ptr, bad.end, nodes[j].desc.kind))
            it.ndeps = len(keep)
            for q, j in enumerate(sorted(keep, key=lambda j: pos[j])):
                it.deps[q] = pos[j]
        d = L.PersistDesc()
        d.kind, d.n, d.blob_bytes = L.OP_PERSIST, len(seg), len(blob)
        hb = C.create_string_buffer(bytes(blob), len(blob))
        d.host_items, d.host_blob = C.addressof(items), C.addressof(hb)
        if any(items[k]._pad[1] < 0 for k in range(len(seg))):
            win = [items[k]._pad[1] for k in range(len(seg))]
            for k in range(len(seg)):
                items[k]._pad[1] = 0
            dry = L.PersistDesc.from_buffer_copy(d)
            L.check(L.load().nasrec_persist_prepare(C.addressof(dry)))  # (geometry only: fills items[k].nblk)
            for k in range(len(seg)):
                if win[k] < 0:
                    units = items[k].nblk
                    items[k]._pad[1] = max(32, min(units, -(-units * 5000 // -win[k])))
                else:
                    items[k]._pad[1] = 0
                items[k].nblk = items[k].first = 0
        d.nodes = [nodes[i] for i in seg]
        d._host = (items, hb)
        if dev_alloc is None:  # (descriptors and geometry only: CPU tests of the packing, tools)
            L.check(L.load().nasrec_persist_prepare(C.addressof(d)))
            out.append(d)
            seg.clear()
            return
        cap = 4096  # chunk entries: 65 536 workgroups
        t_items = dev_alloc(C.sizeof(L.PersistItem) * len(seg))
        t_blob = dev_alloc(len(blob) + 64)  # (the descriptor warm-up reads whole 64-byte steps)
        t_chunk = dev_alloc(2 * cap)
        t_cnt = dev_alloc(8 * len(seg) * (L.PS_SHARDS + 1) * L.PS_COUNTER_STRIDE)
        t_flags = dev_alloc(4 * len(seg) * L.PS_REPL * L.PS_FLAG_STRIDE)
        t_err = dev_alloc(16)
        d.items, d.blob, d.chunk_item, d.counters, d.flags, d.err = (t.data_ptr() for t in (t_items, t_blob, t_chunk, t_cnt, t_flags, t_err))
        d.chunk_cap = cap
        L.check(L.load().nasrec_persist_prepare(C.addressof(d)))
        d._keep = (items, hb, t_items, t_blob, t_chunk, t_cnt, t_flags, t_err)
        d.err_tensor = t_err
        out.append(d)
        seg.clear()

    hint = [0] * n
    if _PS_ORDER == "level":
        order = sorted(range(n), key=lambda i: (nodes[i].level, item_bytes(nodes[i]) is not None, -_cost(nodes[i])))
    else:
        # Workgroups are dispatched in index order, so the item order IS the priority order.  key = ASAP start + alpha * slack on a
        # schedule without resource limits (durations: the cost model minus what a launch of its own would add): an item on the
        # critical path (slack 0) sits at its earliest start; weight-gradient products and second passes nobody waits for slide towards
        # their latest start and fill whatever the chain leaves idle, instead of queueing in front of it.  Any alpha in [0, 1] keeps
        # the order topological (both the ASAP and the ALAP start of a successor exceed its predecessor's by the predecessor's duration).
        dur = [max(_cost_r4(nodes[i]) - 2500, 1200) for i in range(n)]
        asap = [0] * n
        for i in range(n):
            asap[i] = max([asap[j] + dur[j] for j in pred[i]] + [0])
        end = max(asap[i] + dur[i] for i in range(n))
        succ = [[] for _ in range(n)]
        for i in range(n):
            for j in pred[i]:
                succ[j].append(i)
        alap = [0] * n
        for i in range(n - 1, -1, -1):
            alap[i] = min([alap[k] for k in succ[i]] + [end]) - dur[i]
        order = sorted(range(n), key=lambda i: (asap[i] + _PS_ALPHA * (alap[i] - asap[i]), i))
        if _PS_THROTTLE > 0:
            # an operator with slack runs on as few workgroups as finish it inside a fraction of its window (units of ~5 us each)
            for i in range(n):
                slack = alap[i] - asap[i]
                if slack >= 10000:
                    hint[i] = -int(_PS_THROTTLE * slack)  # (negative: nanoseconds of window; turned into a workgroup count once the units are known)
    for i in order:
        if item_bytes(nodes[i]) is None:
            flush()
            assert nodes[i].part == "whole"
            out.append(nodes[i].desc)
        else:
            seg.append(i)
    flush()
    return out, nl


# ----------------------------------------------------------------------------------------------------------------
# dead forward operators
# ----------------------------------------------------------------------------------------------------------------
def eliminate_dead_forward(fwd, later):
    """Forward descriptors whose results nobody reads: not the final logit, not a later forward operator that is itself live, not
    the backward program (`later`: it already holds live branches only — plan.py replays closures with liveness).  A fixed
    sub-network can contain such operators: with last_n_blocks_out = 1 (supernet.py:592-598) the final layer reads the LAST block
    only, and a block no later block selects in its macro choice is computed by the reference and thrown away (in
    ea_criteo_kaggle_xlarge_best_1shot.json that is all of block 5: its 256 x 768 x 1565 Linear — a quarter of the step's flops —
    its Transformer and its dense -> sparse projection).  Dropping them changes no logit, loss, gradient or parameter; the
    reference leaves their parameters at grad None, and so does the engine either way.
    -> (live forward descriptors, dropped descriptors).  Opt-in (SupernetEngine.dead_code_elimination / NASREC_DCE=1): the default
    executes every operator the reference executes."""
    nodes = [Node(d) for d in list(fwd) + list(later)]
    nf = len(fwd)
    live = [False] * nf + [True] * len(later)
    for i in range(nf - 1, -1, -1):
        n = nodes[i]
        if n.writes is None or n.desc.kind == L.OP_FINAL_FWD:
            live[i] = True
            continue
        for j in range(i + 1, len(nodes)):
            if not live[j]:
                continue
            m = nodes[j]
            if m.reads is None or any(overlap(w, r) for w in n.writes for r in m.reads):
                live[i] = True
                break
    return [d for d, k in zip(fwd, live[:nf]) if k], [d for d, k in zip(fwd, live[:nf]) if not k]
