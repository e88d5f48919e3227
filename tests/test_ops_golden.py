"""Pins the oracle's OPERATOR functions (and its block wiring) to per-operator vectors produced by the real reference's
modules (tests/golden/make_golden_ops.py).  CPU-only; the reference itself is not needed."""
import numpy as np
import pytest
import torch

from helpers import proj_checksum
from ops_cases import all_cases, case_inputs, oracle_run

CASES = all_cases()


@pytest.mark.parametrize("case", sorted(CASES), ids=sorted(CASES))
def test_oracle_operator_matches_reference(case):
    z, meta = CASES[case]
    # (1) lazy creation: same key set / shapes / deleted projections as the reference after its warm-up forward
    P, _ = oracle_run(case, z, meta, torch.float64, frozen=False, warm=True)
    assert {k[3:]: list(v.shape) for k, v in P.items()} == meta["param_shapes"]
    # (2) fp64 forward + gradients
    for dtype, tag, tol in ((torch.float64, "f64", 1e-10), (torch.float32, "f32", 3e-5)):
        P, (xs, outs) = oracle_run(case, z, meta, dtype, frozen=True)
        for j, o in enumerate(outs):
            ref = z["%s/out%d_%s" % (case, j, tag)]
            assert tuple(o.shape) == ref.shape
            scale = max(1.0, float(np.abs(ref).max()))
            assert float(np.abs(o.detach().numpy() - ref).max()) <= tol * scale, (case, tag, j)
        if tag != "f64":
            continue
        loss = 0
        from oracle import nasrec_oracle as O
        for j, o in enumerate(outs):
            if o.requires_grad:
                loss = loss + (o * torch.tensor(O.seeded_array("dout:%s:%d" % (case, j), o.shape))).sum()
        grads = torch.autograd.grad(loss, xs + list(P.values()), allow_unused=True) if torch.is_tensor(loss) else [None] * (len(xs) + len(P))
        for i, x in enumerate(xs):
            ref = z["%s/din%d" % (case, i)]
            g = grads[i].numpy() if grads[i] is not None else np.zeros_like(ref)
            assert float(np.abs(g - ref).max()) <= 1e-9 * max(1.0, float(np.abs(ref).max())), (case, "din", i)
        got_none = []
        for (k, p), g in zip(P.items(), grads[len(xs):]):
            name = k[3:]
            if g is None:
                got_none.append(name)
                continue
            dot, nrm = meta["grads"][name]
            d, n = proj_checksum("op." + name, g)
            assert abs(d - dot) <= 1e-9 * max(1.0, nrm) and abs(n - nrm) <= 1e-9 * max(1.0, nrm), (case, name)
        assert sorted(got_none) == sorted(meta["grad_none"])


def test_case_inventory_covers_every_operator_and_mode():
    """fixed + supernet, LayerNorm on/off, relu/silu and the skip-projection corners, per operator class"""
    by_cls = {}
    for case, (z, meta) in CASES.items():
        by_cls.setdefault(meta["cls"], []).append(meta)
    for cls in ("ElasticLinear", "ElasticLinear3D", "DotProduct", "Sum", "SigmoidGating", "Transformer", "FactorizationMachine3D",
                "SuperNetBlock"):
        ms = by_cls[cls]
        assert {m["fixed"] for m in ms} == {True, False}, cls
        lns = {bool(m.get("use_layernorm", m.get("kwargs", {}).get("use_layernorm"))) for m in ms}
        assert lns == {True, False}, cls
    acts = {m["kwargs"]["activation"] for m in by_cls["ElasticLinear"]}
    assert {"relu", "silu", "identity"} <= acts
    assert len(CASES) >= 45
