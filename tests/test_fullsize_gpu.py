"""Parity at BASELINE.json's FULL size (cfg 2: Criteo best-1shot sub-network, batch 256, the full 33.76 M-row tables), where
the fp64 oracle cannot replay whole training runs in seconds: size-independent properties of the path plus an oracle
check of the logits on a slice of the batch.

  * sample independence: logits(batch) == concat(logits(first half), logits(second half)) — the whole network is row-local;
  * embedding stem: the gathered rows are bit-identical to the table rows, including ids 0 and rows-1;
  * oracle: logits of 16 samples against the fp64 restatement run on the engine's own weights (1e-5 bar);
  * row-sparse optimizer == dense-gradient reference semantics: after a step, rows no sample touched are bit-identical and
    their Adagrad state is still 0; every touched row moved;
  * determinism: the same step from the same state gives bit-identical parameters (no float atomics anywhere);
  * hipGraph replay == eager program, bit for bit."""
import json
import os

import pytest
import torch

from nasrec_amd import plan as P
from nasrec_amd.engine import SupernetEngine
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.utils.config import NUM_EMBEDDINGS_CRITEO
from oracle import nasrec_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 256


def _choice():
    ca = json.load(open(os.path.join(ROOT, "nasrec_amd", "configs", "criteo", "ea_criteo_kaggle_xlarge_best_1shot.json")))
    return ca, {"macro": ca["macro"], "micro": ca["micro"]}


@pytest.fixture(scope="module")
def setup():
    ca, choice = _choice()
    cfg = P.NetConfig(ca["num_blocks"], ops_config_lib[ca["config"]], False, "relu", fixed=True)
    eng = SupernetEngine(cfg, 13, 26, NUM_EMBEDDINGS_CRITEO, warm_choice=choice)
    eng.init_weights(seed=0)
    int_x, cat_x, y = O.synthetic_batch(B, 13, NUM_EMBEDDINGS_CRITEO, seed=4321)
    cat_x[0] = 0  # boundary ids
    cat_x[1] = torch.tensor(NUM_EMBEDDINGS_CRITEO) - 1
    cat_x[2] = cat_x[3]  # a fully duplicated sample
    int_x[2] = int_x[3]
    return eng, choice, ca, int_x.cuda(), cat_x.cuda(), y.view(-1).cuda()


def _snapshot(eng):
    torch.cuda.synchronize()
    return (eng.flat_p.clone(), eng.flat_s.clone(), [t.clone() for t in eng.tables], [t.clone() for t in eng.table_state])


def _restore(eng, snap):
    eng.flat_p.copy_(snap[0])
    eng.flat_s.copy_(snap[1])
    for t, s in zip(eng.tables, snap[2]):
        t.copy_(s)
    for t, s in zip(eng.table_state, snap[3]):
        t.copy_(s)
    torch.cuda.synchronize()


def test_sample_independence_and_graph_equals_eager(setup):
    eng, choice, _, int_x, cat_x, _ = setup
    full = eng.forward(int_x, cat_x, choice).clone()
    fullg = eng.forward(int_x, cat_x, choice, graph=True).clone()
    assert torch.equal(full, fullg)  # same launches, replayed
    lo = eng.forward(int_x[:128], cat_x[:128], choice).clone()
    hi = eng.forward(int_x[128:], cat_x[128:], choice).clone()
    halves = torch.cat([lo, hi], 0)
    tol = 1e-5 * max(1.0, float(full.abs().max()))
    assert float((full - halves).abs().max()) <= tol
    assert torch.equal(full[2], full[3])  # identical samples -> identical logits


def test_embedding_stem_bit_exact_on_full_tables(setup):
    eng, choice, _, int_x, cat_x, _ = setup
    eng.forward(int_x, cat_x, choice)
    cp = eng.compile(choice, B, train=False)
    got = cp.sparse0.t.view(B, 26, 16)
    torch.cuda.synchronize()
    for f in range(26):
        assert torch.equal(got[:, f], eng.tables[f][cat_x[:, f]]), f
    eng.check_indices()


def test_logits_against_the_oracle_on_the_engines_weights(setup):
    eng, choice, ca, int_x, cat_x, _ = setup
    n = 16
    # the B = 256 plan (the launches bench.py times), sliced — not a 16-sample plan with other tile / split-K choices; the whole
    # batch incl. the backward is checked in tests/test_operating_point_parity_gpu.py
    logits = eng.forward(int_x, cat_x, choice)[:n].double().cpu().view(-1)
    Pm = O.Params(torch.float64)
    torch.cuda.synchronize()
    for k, v in eng.params.items():
        if not k.startswith("_embedding."):
            Pm[k] = v.detach().double().cpu()
    # only the rows these 16 samples touch are needed: remap ids onto a compact copy of each table
    cat_small = torch.zeros(n, 26, dtype=torch.int64)
    for f in range(26):
        ids, inv = torch.unique(cat_x[:n, f].cpu(), return_inverse=True)
        Pm["_embedding.%d.weight" % f] = eng.tables[f][ids.cuda()].double().cpu()
        cat_small[:, f] = inv
    Pm.frozen = True
    cfg = O.NetCfg(ca["num_blocks"], O.ops_config_lib[ca["config"]], False, "relu", fixed=True)
    ref = O.supernet_forward(Pm, cfg, int_x[:n].double().cpu(), cat_small, choice).view(-1)
    tol = 1e-5 * max(1.0, float(ref.abs().max()))
    assert float((logits - ref).abs().max()) <= tol


def test_rowsparse_step_touches_exactly_the_batch_rows_and_is_deterministic(setup):
    eng, choice, _, int_x, cat_x, y = setup
    lr = 1e-3
    eng.train_step(int_x, cat_x, y, lr, choice=choice)  # allocates the Adagrad state
    snap = _snapshot(eng)
    results = []
    for graph in (False, True, True):
        _restore(eng, snap)
        loss = eng.train_step(int_x, cat_x, y, lr, choice=choice, graph=graph)
        torch.cuda.synchronize()
        results.append((float(loss.item()), eng.flat_p.clone(), [t.clone() for t in eng.tables]))
    for r in results[1:]:
        assert r[0] == results[0][0]
        assert torch.equal(r[1], results[0][1])  # eager == graph == graph again, bit for bit
        for a, b in zip(r[2], results[0][2]):
            assert torch.equal(a, b)
    coef = float(eng.clip_out[0].item())
    assert 0.0 < coef <= 1.0
    after_p, after_tables, after_state = eng.flat_p.clone(), [t.clone() for t in eng.tables], [t.clone() for t in eng.table_state]
    # d loss / d gathered rows of that step: the optimizer sums duplicate rows IN PLACE (csrc/dedup_bodies.h), so the per-sample gradients
    # are taken from a forward + backward of their own on the same (restored) state — the engine is deterministic, these are the same bits
    _restore(eng, snap)
    sg = eng.forward_backward(int_x, cat_x, y, choice).sparse0.grad_tensor().view(B, 26, 16).double().clone()
    torch.cuda.synchronize()
    moved = 0
    for f in range(26):
        before, after = snap[2][f], after_tables[f]
        touched = torch.zeros(before.shape[0], dtype=torch.bool, device=before.device)
        touched[cat_x[:, f]] = True
        changed = (before != after).any(dim=1)
        moved += int(changed.sum())
        assert not bool((changed & ~touched).any()), "table %d: a row outside the batch changed" % f
        st_before, st_after = snap[3][f], after_state[f]
        assert torch.equal(st_before[~touched], st_after[~touched])
        assert torch.equal(before[~touched], after[~touched])
        # the touched rows follow the DENSE reference update (index_add of the per-sample gradients, clip, Adagrad)
        rows = torch.unique(cat_x[:, f])
        g = torch.zeros(before.shape[0] if before.shape[0] < 100000 else 0, 16, dtype=torch.float64, device=before.device)
        if g.shape[0]:
            g.index_add_(0, cat_x[:, f], sg[:, f])
            g = g[rows] * coef
            st = st_before[rows].double() + g * g
            want = before[rows].double() - lr * g / (st.sqrt() + 1e-2)
            assert torch.allclose(after[rows].double(), want, rtol=0, atol=1e-7), "table %d" % f
            assert torch.allclose(st_after[rows].double(), st, rtol=1e-6, atol=1e-12), "table %d" % f
    assert moved > 0 and not torch.equal(snap[0], after_p)


def test_large_evaluation_batch_equals_the_batch_256_path(setup):
    """a 4096-sample forward (the throughput-regime GEMM tiles, the > 256 code paths) reproduces the logits of the same
    samples pushed through the batch-256 plan, 256 at a time"""
    eng, choice, _, _, _, _ = setup
    n = 4096
    int_x, cat_x, _ = O.synthetic_batch(n, 13, NUM_EMBEDDINGS_CRITEO, seed=99)
    int_x, cat_x = int_x.cuda(), cat_x.cuda()
    big = eng.forward(int_x, cat_x, choice).clone()
    small = torch.cat([eng.forward(int_x[i:i + 256], cat_x[i:i + 256], choice, graph=True).clone() for i in range(0, n, 256)], 0)
    tol = 1e-5 * max(1.0, float(small.abs().max()))
    assert float((big - small).abs().max()) <= tol
