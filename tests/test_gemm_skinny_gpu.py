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
