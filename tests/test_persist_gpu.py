"""The persistent step (NASREC_OP_PERSIST, csrc/worklist.hip persist_kernel; nasrec_amd/schedule.py pack_persistent): the worklist items
of many levels in ONE launch, dependencies resolved inside the kernel, the plan's buffers in uncached memory.  The bodies are the level
launches' bodies on the same descriptors, so a persistent step must reproduce the level-scheduled step (and through it the
one-launch-per-operator program: tests/test_parity_gpu.py) BIT FOR BIT — logits, loss, every gradient, parameters after training steps,
launched and replayed as a graph — and must do so every time: the hand-off between workgroups is a protocol (arrival counters, replica
flags, an acquire in front of every body), and a protocol that is wrong is wrong rarely.  Reference step: nasrec/utils/train_utils.py:255-287."""
import os

import pytest
import torch

from helpers import GOLDEN, load_golden
from nasrec_amd import _lib as L
from test_parity_gpu import build_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_scheduler_knobs():
    from nasrec_amd import schedule as S
    keep = (S.RESPLIT, S.RESPLIT_MIN_SLACK)
    yield
    S.RESPLIT, S.RESPLIT_MIN_SLACK = keep


CASES = ["fixed_criteo_xlarge", "fixed_kdd_autoctr", "fixed_criteo_xlarge_ln", "fixed_avazu_xlarge"]


def _inputs(z):
    return torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda(), torch.tensor(z["y"]).cuda().view(-1)


def _run(z, meta, persist, graph, steps=3):
    from nasrec_amd import schedule as S
    int_x, cat_x, y = _inputs(z)
    eng = build_engine(z, meta)
    eng.persist = persist
    # the persistent form re-cuts stand-alone products into split-K items (schedule.PERSIST_RESPLIT: another summation order than the
    # one-pass kernel's): the level-launch baseline runs the same operator list — the same re-cut (NASREC_WL_RESPLIT, same slack rule)
    S.RESPLIT = S.PERSIST_RESPLIT != "off"
    S.RESPLIT_MIN_SLACK = 0 if S.PERSIST_RESPLIT == "all" else 3
    cp = eng.forward_backward(int_x, cat_x, y, meta["choice"])
    torch.cuda.synchronize()
    out = dict(logits=cp.logits.clone(), grads=eng.flat_g.clone(), sg=cp.sparse0.grad_tensor().clone(), loss=float(cp.loss.item()),
               persistent=bool(getattr(cp, "persistent", False)),
               launches=len(cp.fb.descs), n_persist=sum(isinstance(d, L.PersistDesc) for d in cp.fb.descs),
               items=sum(d.n for d in cp.fb.descs if isinstance(d, L.PersistDesc)))
    losses = []
    for _ in range(steps):
        losses.append(float(eng.train_step(int_x, cat_x, y, 0.05, meta["choice"], graph=graph)))
    torch.cuda.synchronize()
    eng.check_indices()  # (also raises when a wait of the persistent kernel ran out of its budget)
    out.update(losses=losses, params=eng.flat_p.clone(), tables=[t.clone() for t in eng.tables])
    return out


@pytest.mark.parametrize("case", CASES)
def test_persistent_step_is_bit_identical_to_the_level_scheduled_step(case):
    z, meta = load_golden(os.path.join(GOLDEN, case + ".npz"))
    base = _run(z, meta, persist=False, graph=False)
    assert not base["persistent"] and base["n_persist"] == 0
    for graph in (False, True):
        got = _run(z, meta, persist=True, graph=graph)
        assert got["persistent"] and got["n_persist"] >= 1, "the plan should have taken the persistent form"
        assert got["launches"] < base["launches"], (got["launches"], base["launches"])
        assert torch.equal(got["logits"], base["logits"]), "logits"
        assert got["loss"] == base["loss"], "loss"
        assert torch.equal(got["grads"], base["grads"]), "dense gradients"
        assert torch.equal(got["sg"], base["sg"]), "row gradients"
        assert got["losses"] == base["losses"], "losses of the training steps"
        assert torch.equal(got["params"], base["params"]), "dense parameters after the steps"
        for a, b in zip(got["tables"], base["tables"]):
            assert torch.equal(a, b)


def test_persistent_step_gives_the_same_bits_every_time():
    """400 consecutive steps on changing batches, persistent against level launches, compared after every 50: a stale read or a missed
    dependency shows up as a differing bit sooner or later (the seam probe's forms were validated the same way: tools/micro/seam_probe.hip)"""
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_criteo_xlarge.npz"))
    int_x, cat_x, y = _inputs(z)
    g = torch.Generator(device="cpu").manual_seed(5)
    B = int_x.shape[0]
    tabs = torch.tensor(meta["tables"])
    batches = []
    for _ in range(16):
        ix = (int_x + 0.1 * torch.randn(int_x.shape, generator=g).cuda()).contiguous()
        cx = (torch.randint(0, 1 << 30, cat_x.shape, generator=g) % tabs[None, :]).cuda()
        yy = (torch.rand(B, generator=g) < 0.3).float().cuda()
        batches.append((ix, cx, yy))
    from nasrec_amd import schedule as S
    S.RESPLIT = S.PERSIST_RESPLIT != "off"  # (the baseline's operator list = the persistent form's: see _run)
    S.RESPLIT_MIN_SLACK = 0 if S.PERSIST_RESPLIT == "all" else 3
    engs = []
    for persist in (False, True):
        e = build_engine(z, meta)
        e.persist = persist
        engs.append(e)
    for step in range(400):
        bx = batches[step % len(batches)]
        la = engs[0].train_step(bx[0], bx[1], bx[2], 0.01, meta["choice"], graph=False)
        lb = engs[1].train_step(bx[0], bx[1], bx[2], 0.01, meta["choice"], graph=(step >= 200))
        if step % 50 == 49:
            torch.cuda.synchronize()
            assert float(la) == float(lb), (step, float(la), float(lb))
            assert torch.equal(engs[0].flat_p, engs[1].flat_p), "parameters differ after step %d" % step
    engs[1].check_indices()
    for a, b in zip(engs[0].tables, engs[1].tables):
        assert torch.equal(a, b)
