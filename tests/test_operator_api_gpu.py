"""The stand-alone operator API on the HIP engine against the per-operator vectors of the REAL reference
(tests/golden/ops_*.npz): every choice-block operator and SuperNetBlock, fixed and supernet (masked) mode,
LayerNorm on/off, relu/silu/identity, and the skip-projection corners.  Outputs within 1e-5 * max(1, |out|_max) of the
reference's fp64 evaluation, input gradients and parameter-gradient checksums within 2e-5 of the gradient norm, identical
`grad is None` sets.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

from helpers import proj_checksum
from ops_cases import all_cases, case_inputs
from ops_modules import build_module
from oracle import nasrec_oracle as O

pytestmark = pytest.mark.gpu
CASES = all_cases()


def _call(m, meta, xs):
    if meta["cls"] == "SuperNetBlock":
        return list(m(xs, meta["choice"]))
    return [m(*xs, meta["dims_in_use"])]


@pytest.mark.parametrize("case", sorted(CASES), ids=sorted(CASES))
def test_operator_matches_reference(case):
    z, meta = CASES[case]
    m = build_module(meta).cuda()
    xs = case_inputs(case, z, meta, torch.float32, device="cuda", requires_grad=False)
    with torch.no_grad():
        _call(m, meta, xs)  # first forward: lazy shapes, deleted projections
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == meta["param_shapes"]
    m.load_state_dict({k: torch.tensor(O.seeded_param("op." + k, s), dtype=torch.float32) for k, s in meta["param_shapes"].items()}, strict=True)
    m.train()
    xs = case_inputs(case, z, meta, torch.float32, device="cuda", requires_grad=True)
    outs = _call(m, meta, xs)
    loss = 0
    for j, o in enumerate(outs):
        ref = z["%s/out%d_f64" % (case, j)]
        assert tuple(o.shape) == ref.shape
        scale = max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(o.detach().cpu().numpy().astype(np.float64) - ref).max())
        assert err <= 1e-5 * scale, (case, "out", j, err, scale)
        if o.requires_grad:
            loss = loss + (o * torch.tensor(O.seeded_array("dout:%s:%d" % (case, j), o.shape), dtype=torch.float32, device="cuda")).sum()
    if not torch.is_tensor(loss):
        assert meta["cls"].startswith("Zeros")
        return
    loss.backward()
    for i, x in enumerate(xs):
        ref = z["%s/din%d" % (case, i)]
        g = x.grad.cpu().numpy().astype(np.float64) if x.grad is not None else np.zeros_like(ref)
        tol = 2e-5 * max(float(np.linalg.norm(ref)), 1e-6)
        assert float(np.abs(g - ref).max()) <= tol, (case, "din", i, float(np.abs(g - ref).max()), tol)
    none = sorted(n for n, p in m.named_parameters() if p.grad is None)
    assert none == sorted(meta["grad_none"])
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        dot, nrm = meta["grads"][n]
        d, nn_ = proj_checksum("op." + n, p.grad)
        tol = 2e-5 * max(nrm, 1e-6)
        assert abs(nn_ - nrm) <= tol and abs(d - dot) <= tol, (case, n, d - dot, nn_ - nrm, tol)


def test_no_grad_forward_equals_grad_forward_and_second_forward_guards_backward():
    from nasrec_amd.supernet.modules import SigmoidGating
    torch.manual_seed(0)
    m = SigmoidGating(fixed=False, use_layernorm=True, max_dims_or_dims=64).cuda()
    l, r = torch.randn(8, 24, device="cuda"), torch.randn(8, 40, device="cuda")
    with torch.no_grad():
        a = m(l, r, 32)
    b = m(l, r, 32)
    assert torch.equal(a, b.detach())
    assert float(a[:, 32:].abs().max()) == 0.0  # prefix mask (modules.py:589-592)
    c = m(l, r, 32)  # same plan again before b's backward: b's saved activations are gone
    with pytest.raises(RuntimeError):
        b.sum().backward()
    c.sum().backward()


def test_operator_inside_a_bound_supernet_is_callable_on_its_own():
    """a node of a materialised SuperNet (parameters living in the engine's flat arena) runs stand-alone on the same weights"""
    import os
    from helpers import GOLDEN, load_golden
    from nasrec_amd.search_space import ops_config_lib
    from nasrec_amd.supernet.supernet import SuperNet
    z, meta = load_golden(os.path.join(GOLDEN, "fixed_criteo_xlarge.npz"))
    model = SuperNet(num_blocks=meta["num_blocks"], ops_config=ops_config_lib[meta["config"]], use_layernorm=False, num_embeddings=meta["tables"],
                     sparse_input_size=26, path_sampling_strategy="fixed-path", fixed=True, fixed_choice=meta["choice"]).cuda()
    int_x, cat_x = torch.tensor(z["int_x"]).cuda(), torch.tensor(z["cat_x"]).cuda()
    with torch.no_grad():
        model(int_x, cat_x)
        node = model._blocks[0]._nodes[2]  # SigmoidGating 13 -> 128 (cfg 1, block 0)
        out = node(int_x, int_x, 128)
        w, b = node._left_self_linear._linear.weight, node._left_self_linear._linear.bias
        g = torch.sigmoid(int_x.double() @ w.double().t() + b.double()) * int_x.double()
        want = g @ node._linear_proj.weight.double().t() + node._linear_proj.bias.double()
    assert float((out.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


def test_silu_and_sigmoid_epilogues_over_the_whole_range():
    """The GEMM epilogues evaluate SiLU / sigmoid with the hardware exponential (`__expf`, csrc/common.h).  Pin them against fp64
    over pre-activations from -100 to 100 (saturation, the subnormal tail of exp(-z), the steep middle), through the operator API
    with identity weights.  exp(-z) is evaluated as exp2(-z * log2 e) in fp32, so its relative error grows like |z| * 6e-8: the
    bar is 4 ulp * (1 + |z| / 8) of the RESULT — at most 6e-6 relative, and only where the result itself is below 1e-30 in
    magnitude; absolute errors stay below 1e-6 * |result| everywhere that matters for the 1e-5 logit bar."""
    from nasrec_amd.supernet.modules import ElasticLinear, SigmoidGating
    z = torch.cat([torch.linspace(-100, 100, 4001), torch.tensor([-87.4, -20.0, -1e-3, 0.0, 1e-3, 16.6, 88.8])]).double()
    n = (z.numel() + 15) // 16 * 16
    z = torch.cat([z, torch.zeros(n - z.numel(), dtype=torch.float64)]).view(-1, 16)
    x = z.float().cuda()
    lin = ElasticLinear(fixed=True, use_layernorm=False, max_dims_or_dims=16, activation="silu").cuda()
    with torch.no_grad():
        lin(x, 16)
        lin._linear.weight.copy_(torch.eye(16))
        lin._linear.bias.zero_()
        silu = lin(x, 16).double().cpu()
    zz = x.double().cpu()
    ref = zz / (1.0 + torch.exp(-zz))
    err = (silu - ref).abs()
    assert bool((err <= 4 * 1.2e-7 * (1 + zz.abs() / 8) * ref.abs() + 1e-30).all()), float((err / (ref.abs() + 1e-30)).max())
    gate = SigmoidGating(fixed=True, use_layernorm=False, max_dims_or_dims=16).cuda()
    ones = torch.ones_like(x)
    with torch.no_grad():
        gate(x, ones, 16)
        gate._left_self_linear._linear.weight.copy_(torch.eye(16))
        gate._left_self_linear._linear.bias.zero_()
        sig = gate(x, ones, 16).double().cpu()
    ref = 1.0 / (1.0 + torch.exp(-zz))
    err = (sig - ref).abs()
    assert bool((err <= 4 * 1.2e-7 * (1 + zz.abs() / 8) * ref.abs() + 1e-38).all()), float((err / (ref.abs() + 1e-38)).max())
