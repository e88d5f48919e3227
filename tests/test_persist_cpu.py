"""CPU tests of the persistent form's packing (nasrec_amd/schedule.py pack_persistent; csrc/worklist.hip nasrec_persist_prepare in its
geometry-only mode): descriptors and workgroup ranges only, nothing is launched.  Checked: every dependency the footprints imply is
ordered by a path of item dependencies inside a launch (or by a launch boundary), items only ever wait for EARLIER items (the forward-
progress argument of the kernel: workgroups are dispatched in index order), the dependency lists fit, the workgroup ranges tile the
launch, the footprint audit refuses a program whose hand-offs run through cached memory, and the throttling hints cap the bulk operators."""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from nasrec_amd import _lib as L  # noqa: E402
from nasrec_amd import schedule as S  # noqa: E402

CFG = os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")
EVERYWHERE = [(0, 1 << 62)]


@pytest.fixture(scope="module")
def descs():
    import show_levels as SL
    ctx = SL.build_cpu_plan(CFG, 256, 13, 26)
    return list(ctx.fwd) + list(ctx.bwd)


@pytest.mark.parametrize("order,throttle", [("start", 0.5), ("start", 0.0), ("level", 0.0)])
def test_packing_orders_every_dependency_and_tiles_the_launch(descs, order, throttle, monkeypatch):
    monkeypatch.setattr(S, "_PS_ORDER", order)
    monkeypatch.setattr(S, "_PS_THROTTLE", throttle)
    out, nl = S.pack_persistent(descs, None, EVERYWHERE)
    assert any(isinstance(d, L.PersistDesc) for d in out) and len(out) < nl
    placed = {}  # node -> (launch index, item index)
    for li, d in enumerate(out):
        if not isinstance(d, L.PersistDesc):
            continue
        items = d._host[0]
        first = 0
        for k in range(d.n):
            it = items[k]
            placed[id(d.nodes[k])] = (li, k)
            assert it.first == first and it.nblk >= 1 and 1 <= it._pad[1] <= it.nblk, (k, it.first, first, it.nblk, it._pad[1])
            first += it._pad[1]
            assert 0 <= it.ndeps <= L.PS_MAX_DEPS
            assert all(0 <= it.deps[q] < k for q in range(it.ndeps)), "an item may only wait for earlier items"
        assert first == d.total_blocks
        # reachability through the (transitively reduced) item dependencies
        reach = [set() for _ in range(d.n)]
        for k in range(d.n):
            for q in range(items[k].ndeps):
                j = items[k].deps[q]
                reach[k] |= {j} | reach[j]
        for k in range(d.n):
            for j in range(k):
                if S._depends(d.nodes[k], d.nodes[j]) or S._depends(d.nodes[j], d.nodes[k]):
                    assert j in reach[k], "launch %d: item %d depends on item %d without a path of waits" % (li, k, j)
    # across launches (and stand-alone kernels between them) program order is the dependency order
    seq = []
    for li, d in enumerate(out):
        seq += [(li, n) for n in d.nodes] if isinstance(d, L.PersistDesc) else [(li, S.Node(d))]
    for a in range(len(seq)):
        for b in range(a):
            if seq[a][0] != seq[b][0] and S._depends(seq[b][1], seq[a][1]) and not S._depends(seq[a][1], seq[b][1]):
                # (an earlier launch must not depend on a later one; _depends is symmetric in its hazards, so this can only be a pure
                # read-after-write the wrong way round — checked through the producers' writes)
                assert not any(S.overlap(w, r) for w in seq[a][1].writes for r in seq[b][1].reads), (seq[b][0], seq[a][0])
    if throttle > 0:
        caps = [d._host[0][k]._pad[1] < d._host[0][k].nblk for d in out if isinstance(d, L.PersistDesc) for k in range(d.n)]
        assert any(caps), "no operator with slack was throttled"


def test_hand_offs_through_cached_memory_are_refused(descs):
    with pytest.raises(S.PersistRefused):
        S.pack_persistent(descs, None, [(0, 4096)])  # (no buffer of the plan lies in this "arena")


def test_item_records_match_the_c_structs():
    lib = L.load()
    sizes = (C.c_int32 * 40)()
    n = lib.nasrec_desc_sizes(sizes, 40)
    assert n > 37 and sizes[L.OP_PERSIST] == C.sizeof(L.PersistDesc) and sizes[37] == C.sizeof(L.PersistItem) == 80
