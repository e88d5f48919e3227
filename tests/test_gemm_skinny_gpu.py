"""The narrow-output forward Linear at large batch (csrc/gemm_skinny.hip: N <= 16, x streamed once, K split inside the workgroup) through
the C-ABI against fp64 products: K-segmented inputs with unaligned row strides and ragged tails (a 3-wide and a 13-wide segment, a dead
segment in the middle), ragged M, N = 16 / 13 / 1, both wavefront counts (M / 16 below and above 512 row tiles), the common epilogue
(bias, ReLU, saved pre-activation, accumulation into the output), and run-to-run determinism."""
import ctypes as C

import pytest
import torch

from nasrec_amd import _lib as L
from nasrec_amd import plan as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return L.load()


def _rand(*shape, scale=1.0):
    return (torch.randn(*shape, device="cuda") * scale).contiguous()


def _close(got, want, tol=2e-5):
    scale = max(1.0, float(want.abs().max()))
    err = float((got.double() - want).abs().max())
    assert err <= tol * scale, "max err %.3e (scale %.3e)" % (err, scale)


@pytest.mark.parametrize("M,N,Ks,epilogue", [
    (8200, 16, [3, 1024, 0, 1024, 1000], "bias_relu_savez"),   # 513 row tiles: four wavefronts per workgroup; a dead segment
    (4100, 16, [13, 1024, 1037], "accumulate"),               # 257 row tiles: eight wavefronts; ragged last tile (4 rows)
    (4096, 13, [300], "plain"),
    (2048, 1, [1024, 16], "bias"),
])
def test_narrow_forward_product_against_fp64(lib, M, N, Ks, epilogue):
    torch.manual_seed(5)
    xs = [(_rand(M, k + 3)[:, :k] if k else None) for k in Ks]      # row strides that are not multiples of 4 floats
    Ws = [(_rand(N, k + 1, scale=0.05)[:, :k] if k else None) for k in Ks]
    out = torch.full((M, N + 3), 7.0, device="cuda")
    z = torch.zeros(M, N + 3, device="cuda")
    bias = _rand(N)
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.splitk, d.dims_in_use = L.OP_GEMM, L.AM_KC, L.AM_KC, L.CM_PLAIN, len(Ks), 0, 1, -1
    for q, (x, W, k) in enumerate(zip(xs, Ws, Ks)):
        s = d.seg[q]
        s.A, s.B = (x.data_ptr(), W.data_ptr()) if k else (None, None)
        s.C, s.M, s.N, s.K, s.Mvalid = out.data_ptr(), M, N, k, M
        s.lda, s.ldb, s.ldc = (x.stride(0), W.stride(0), N + 3) if k else (0, 0, N + 3)
    if "bias" in epilogue:
        d.bias = bias.data_ptr()
    if "relu" in epilogue:
        d.act = L.ACT_RELU
    if "savez" in epilogue:
        d.save_z = z.data_ptr()
    if epilogue == "accumulate":
        d.beta = 1
    assert P.gemm_kernel_name(d) == "gemm_skinny_n_kernel", "the case must be sized for the narrow-output kernel"
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()
    first = out.clone()
    zz = sum(x.double() @ W.double().t() for x, W, k in zip(xs, Ws, Ks) if k)
    if "bias" in epilogue:
        zz = zz + bias.double()
    want = torch.relu(zz) if "relu" in epilogue else zz
    if epilogue == "accumulate":
        want = want + 7.0
    _close(out[:, :N], want)
    if "savez" in epilogue:
        _close(z[:, :N], zz)
    assert bool((out[:, N:] == 7.0).all()), "columns beyond N were written"
    # the same launch again on the same inputs: the same bits (fixed summation order)
    out.fill_(7.0)
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()
    assert torch.equal(out, first)


@pytest.mark.parametrize("M,N,K,acc", [(8192, 16, 1024, 1), (4100, 13, 1000, 0), (2048, 3, 300, 1)])
def test_narrow_input_gradient_against_fp64(lib, M, N, K, acc):
    """dx[M, N] = dy[M, K] W[K, N] (binding KC / RC, one problem of a zmode launch): the input gradient of a Linear with <= 16 inputs,
    with and without accumulation into a gradient that already holds a contribution"""
    torch.manual_seed(6)
    dy = _rand(M, K + 1)[:, :K]
    W = _rand(K, N + 2, scale=0.05)[:, :N]
    dx = torch.full((M, N + 1), 0.5, device="cuda")
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.splitk, d.dims_in_use = L.OP_GEMM, L.AM_KC, L.AM_RC, L.CM_PLAIN, 1, 1, 1, -1
    s = d.seg[0]
    s.A, s.B, s.C, s.M, s.N, s.K, s.Mvalid = dy.data_ptr(), W.data_ptr(), dx.data_ptr(), M, N, K, M
    s.lda, s.ldb, s.ldc, s.accumulate = dy.stride(0), W.stride(0), N + 1, acc
    assert P.gemm_kernel_name(d) == "gemm_skinny_n_kernel"
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()
    want = dy.double() @ W.double() + (0.5 if acc else 0.0)
    _close(dx[:, :N], want)
    assert bool((dx[:, N:] == 0.5).all()), "columns beyond N were written"


def test_plan_gives_the_narrow_product_one_pass():
    """plan.gemm_descs mirrors the kernel's rule: no split-K workspace for a launch the narrow-output kernel takes"""
    class Ctx:
        B = 4096
        sk_workspace = None
        shape_only = False

        def alloc(self, n):
            return torch.empty(int(n), device="cuda")
    x, W, out = _rand(4096, 1024), _rand(16, 1024), torch.zeros(4096, 16, device="cuda")
    sd = dict(A=x.data_ptr(), B=W.data_ptr(), C=out.data_ptr(), M=4096, N=16, K=1024, lda=1024, ldb=1024, ldc=16)
    d = P.gemm_descs(Ctx(), L.AM_KC, L.AM_KC, L.CM_PLAIN, [sd], 0)[0]
    assert d.splitk == 1 and not d.workspace and P.gemm_kernel_name(d) == "gemm_skinny_n_kernel"



@pytest.mark.parametrize("M,N,K,epilogue", [(4096, 1024, 13, "bias_relu_savez"), (4100, 1000, 16, "accumulate"), (2050, 300, 5, "bias"), (1024, 256, 1, "plain")])
def test_wide_output_from_few_inputs_against_fp64(lib, M, N, K, epilogue):
    """round 6, csrc/gemm_skinny.hip gemm_tinyk_kernel: y[M, N] = x[M, K] W[N, K]^T with K <= 16 at large batch (the projections of the raw dense
    features: supernet.py:1137-1145, modules.py:171) — a streaming write, weights in registers"""
    torch.manual_seed(8)
    x = _rand(M, K + 3)[:, :K]
    W = _rand(N, K + 1, scale=0.3)[:, :K]
    out = torch.full((M, N + 3), 7.0, device="cuda")
    z = torch.zeros(M, N + 3, device="cuda")
    bias = _rand(N)
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.splitk, d.dims_in_use = L.OP_GEMM, L.AM_KC, L.AM_KC, L.CM_PLAIN, 1, 0, 1, -1
    s0 = d.seg[0]
    s0.A, s0.B, s0.C, s0.M, s0.N, s0.K, s0.Mvalid = x.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, M
    s0.lda, s0.ldb, s0.ldc = x.stride(0), W.stride(0), N + 3
    if "bias" in epilogue:
        d.bias = bias.data_ptr()
    if "relu" in epilogue:
        d.act = L.ACT_RELU
    if "savez" in epilogue:
        d.save_z = z.data_ptr()
    if epilogue == "accumulate":
        d.beta = 1
    assert P.gemm_kernel_name(d) == "gemm_tinyk_kernel"
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()
    zz = x.double() @ W.double().t()
    if "bias" in epilogue:
        zz = zz + bias.double()
    want = torch.relu(zz) if "relu" in epilogue else zz
    if epilogue == "accumulate":
        want = want + 7.0
    _close(out[:, :N], want)
    if "savez" in epilogue:
        _close(z[:, :N], zz)
    assert bool((out[:, N:] == 7.0).all()), "columns beyond N were written"


@pytest.mark.skipif(__import__("os").environ.get("NASREC_TINYK") != "2", reason="the KC / RC form of the tiny-K kernel is an A/B variant (NASREC_TINYK=2): measured no faster than the throughput tile")
def test_input_gradients_of_narrow_linears_as_one_streaming_launch(lib):
    """... and dx_q = dy_q W_q (KC / RC, one problem per segment of a zmode launch, K <= 16 each): three problems of different K and N,
    one accumulating into a gradient that already holds a contribution, one dead (no operand: its output is zeroed)"""
    torch.manual_seed(9)
    M = 4100
    Ks, Ns, accs = [16, 9, 16, 13], [1024, 300, 1024, 256], [0, 1, 1, 0]
    dys = [_rand(M, k + 2)[:, :k] for k in Ks]
    Ws = [_rand(k, n + 1, scale=0.2)[:, :n] for k, n in zip(Ks, Ns)]
    dxs = [torch.full((M, n + 2), 0.25, device="cuda") for n in Ns]
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.splitk, d.dims_in_use = L.OP_GEMM, L.AM_KC, L.AM_RC, L.CM_PLAIN, 4, 1, 1, -1
    for q in range(4):
        s = d.seg[q]
        dead = q == 3
        s.A, s.B = (None, None) if dead else (dys[q].data_ptr(), Ws[q].data_ptr())
        s.C, s.M, s.N, s.K, s.Mvalid = dxs[q].data_ptr(), M, Ns[q], Ks[q], M
        s.lda, s.ldb, s.ldc, s.accumulate = dys[q].stride(0), Ws[q].stride(0), Ns[q] + 2, accs[q]
    assert P.gemm_kernel_name(d) == "gemm_tinyk_kernel"
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()
    for q in range(3):
        want = dys[q].double() @ Ws[q].double() + (0.25 if accs[q] else 0.0)
        _close(dxs[q][:, :Ns[q]], want)
        assert bool((dxs[q][:, Ns[q]:] == 0.25).all())
    assert bool((dxs[3][:, :Ns[3]] == 0.0).all()) and bool((dxs[3][:, Ns[3]:] == 0.25).all()), "a dead problem zeroes its output"
