"""CPU tests of the level scheduler (nasrec_amd/schedule.py) and of the data-parallel step's segmentation helpers
(nasrec_amd/parallel.py) on the launch plan of the bench network (Criteo best-1shot sub-network, batch 256), built without a GPU
(descriptors only, tools/show_levels.build_cpu_plan): nothing is launched here — what is checked is that the schedules are VALID
(every dependency the footprints imply is respected, launches stay within the worklist kernel's item / descriptor limits) and that the
gradient pieces of the exchange are final when they are sent."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from nasrec_amd import _lib as L  # noqa: E402
from nasrec_amd import schedule as S  # noqa: E402

CFG = os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")


@pytest.fixture(scope="module")
def plan():
    import show_levels as SL
    return SL.build_cpu_plan(CFG, 256, 13, 26)


def _check_valid(nodes):
    for i, a in enumerate(nodes):
        for j in range(i):
            if S._depends(a, nodes[j]):
                assert a.level > nodes[j].level, (i, j, a.level, nodes[j].level)


@pytest.mark.parametrize("push", [True, False])
def test_balanced_levels_respect_every_dependency_and_launch_limits(plan, push, monkeypatch):
    monkeypatch.setattr(S, "_PUSH", push)
    descs = list(plan.fwd) + list(plan.bwd)
    nodes = S.expand_for_worklists(descs)
    nl = S.assign_levels(nodes)
    asap = [n.level for n in nodes]
    _check_valid(nodes)

    def total():
        t = 0
        for lv in range(nl):
            mem = [n for n in nodes if n.level == lv]
            t += S._level_ns([n for n in mem if S.item_bytes(n) is not None]) + sum(S._cost(n) for n in mem if S.item_bytes(n) is None)
        return t
    before = total()
    S.balance_levels(nodes, nl)
    _check_valid(nodes)
    assert max(n.level for n in nodes) == nl - 1 and min(n.level for n in nodes) == 0  # the number of levels does not grow
    assert total() < before  # the estimate the pass minimises went down ...
    assert any(a != n.level for a, n in zip(asap, nodes))  # ... because something moved
    for n, a in zip(nodes, asap):
        if S.item_bytes(n) is None:
            assert n.level == a  # launches of their own stay where they are
    for lv in range(nl):
        mem = [n for n in nodes if n.level == lv and S.item_bytes(n) is not None]
        assert len(mem) <= L.WL_MAX_ITEMS and sum((len(S.item_bytes(n)) + 15) & ~15 for n in mem) <= L.WL_BLOB_BYTES


def test_pack_emits_one_worklist_per_level_and_keeps_every_operator(plan):
    descs = list(plan.fwd) + list(plan.bwd)
    out, nl = S.pack(descs)
    seen = []
    for d in out:
        if isinstance(d, L.WorklistDesc):
            assert 1 <= d.n <= L.WL_MAX_ITEMS and len(d.nodes) == d.n
            seen += [(id(n.desc), n.part) for n in d.nodes]
        else:
            seen.append((id(d), "whole"))
    want = []
    for n in S.expand_for_worklists(descs):
        want.append((id(n.desc), n.part))
    assert sorted(seen) == sorted(want)
    assert len(out) <= nl + 4  # a level is one worklist launch plus the few operators that keep a kernel of their own


class _FakeEngine:
    """flat gradient arena + offsets, as parallel.gradient_ready_index / merge_over_unwritten read them"""

    def __init__(self, sizes):
        self.offsets, off = {}, 0
        for k, n in sizes.items():
            self.offsets[k] = off
            off += (n + 3) // 4 * 4
        self.flat_numel = off
        self.flat_g = torch.zeros(off)
        self.params = {k: self.flat_g[o:o + sizes[k]] for k, o in self.offsets.items()}


def test_gradient_pieces_are_cut_where_gradients_are_final_and_merged_only_over_unwritten_ones():
    from nasrec_amd.parallel import cut_segments, gradient_ready_index, merge_over_unwritten
    eng = _FakeEngine({"a": 600, "dead": 100, "b": 10, "c": 10, "d": 8, "dead2": 50, "e": 6, "f": 6})
    base = eng.flat_g.data_ptr()

    def writer(*names):  # a row-reduction launch writing the gradients of `names`
        r = L.ReduceRowsDesc()
        r.kind, r.R, r.C, r.ld, r.in_, r.ndst = L.OP_REDUCE_ROWS, 4, 16, 16, base + 4 * eng.flat_numel + 4096, len(names)
        for q, n in enumerate(names):
            r.dst[q], r.dst_off[q], r.dst_len[q] = base + 4 * eng.offsets[n], 0, eng.params[n].numel()
        return r
    descs = [writer("a"), writer("b"), writer("a"), writer("c", "b"), writer("d"), writer("e"), writer("f")]
    ready = gradient_ready_index(eng, descs)
    assert ready == {"a": 2, "b": 3, "c": 3, "d": 4, "e": 5, "f": 6}  # last writer; "dead" / "dead2" are never written
    numel = {k: eng.params[k].numel() for k in ready}
    pieces = cut_segments(ready, len(descs), numel, 4)
    assert [e for e, _ in pieces] == sorted(e for e, _ in pieces) and pieces[-1][0] == len(descs)
    sent = [n for _, names in pieces for n in names]
    assert sorted(sent) == sorted(ready)  # every written gradient travels exactly once
    for end, names in pieces:
        assert all(ready[n] < end for n in names)  # ... after its last writer
    assert len(pieces) >= 3  # one huge parameter does not use up the cuts
    order = sorted(eng.offsets, key=lambda n: eng.offsets[n])
    assert merge_over_unwritten(eng, order, {"a", "b", "c"}, ready) == [(0, eng.offsets["c"] + 10)]  # across the unwritten "dead"
    assert merge_over_unwritten(eng, order, {"a", "c"}, ready) == [(0, 600), (eng.offsets["c"], 10)]  # never across "b" (another piece's)
    assert merge_over_unwritten(eng, order, {"d", "e"}, ready) == [(eng.offsets["d"], eng.offsets["e"] + 6 - eng.offsets["d"])]
