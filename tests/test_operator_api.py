"""Host logic of the stand-alone operator API (no GPU): for every per-operator golden case the module tree after the first
forward's shape fixing — state_dict keys, shapes, parameters() order, deleted projections — equals the reference's
(recorded in tests/golden/ops_*.npz by importing its modules)."""
import pytest
import torch

from ops_cases import all_cases
from ops_modules import build_module

CASES = all_cases()


@pytest.mark.parametrize("case", sorted(CASES), ids=sorted(CASES))
def test_lazy_shapes_and_deleted_projections_match_reference(case):
    z, meta = CASES[case]
    m = build_module(meta)
    if meta["cls"] == "SuperNetBlock":
        m.materialize(meta["in_shapes"], meta["choice"])
    elif hasattr(m, "materialize"):
        m.materialize(meta["in_shapes"], meta["dims_in_use"])
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == meta["param_shapes"]
    assert [n for n, _ in m.named_parameters()] == meta["param_order"]


def test_operator_on_cpu_tensors_fails_loudly():
    from nasrec_amd._lib import EngineError
    from nasrec_amd.supernet.modules import ElasticLinear
    m = ElasticLinear(fixed=True, use_layernorm=False, max_dims_or_dims=8, activation="relu")
    with pytest.raises((EngineError, OSError)):
        m(torch.zeros(2, 5), 8)
