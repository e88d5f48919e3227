"""The single-pass batch-256 forward GEMM (csrc/gemm_kslice.hip: K split inside a 1024-thread workgroup, LDS-DMA staging, feeder /
multiplier waves) through the C-ABI against fp64 products: y = act(x W^T + b) for K-segmented inputs (nn.Linear over a concatenation,
modules.py:171,489,515,584) — output shapes that are not multiples of the 32 x 32 tile, K-segments that end inside a 128-deep chunk or
inside a 16-byte piece, the tiny segments (<= 16) that bypass the chunk sequence, a dead segment, rows that start on 4-byte boundaries
only, the saved pre-activation, accumulation into the output; and that repeated launches are bit-identical."""
import ctypes as C

import pytest
import torch

from nasrec_amd import _lib as L
from nasrec_amd import plan as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return L.load()


CASES = [
    # (M, N, segment widths (0 = dead segment), bias, act, save_z, beta)
    (256, 768, [13, 768, 768, 16], True, L.ACT_RELU, False, False),   # the dominant launch of the Criteo best-1shot step
    (256, 768, [780], False, L.ACT_NONE, False, False),
    (250, 700, [100, 421, 5], True, L.ACT_NONE, False, False),        # ragged tiles, a 5-wide tiny segment, K = 526
    (131, 1000, [515], False, L.ACT_RELU, True, False),               # 5 x 32 tiles, a chunk tail of 3, saved pre-activation
    (256, 768, [300, 0, 301], True, L.ACT_NONE, False, True),         # a dead segment, accumulation into y
    (200, 650, [16, 16, 129, 500], True, L.ACT_SILU, False, False),   # two tiny segments + a tail of 1
    (256, 512, [1027], False, L.ACT_NONE, False, False),              # 128 tiles: the smallest eligible grid
]


@pytest.mark.parametrize("M,N,widths,bias,act,save_z,beta", CASES)
def test_kslice_matches_fp64(lib, M, N, widths, bias, act, save_z, beta):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    K = sum(widths)
    xs = [torch.randn(M, max(wd, 1), device="cuda", generator=g) if wd else None for wd in widths]
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g) if bias else None
    y = torch.randn(M, N, device="cuda", generator=g)
    y0 = y.clone()
    z = torch.empty(M, N, device="cuda") if save_z else None
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use, d.splitk = L.OP_GEMM, L.AM_KC, L.AM_KC, L.CM_PLAIN, len(widths), 0, -1, 1
    d.act, d.beta = act, 1 if beta else 0
    if b is not None:
        d.bias = b.data_ptr()
    if z is not None:
        d.save_z = z.data_ptr()
    off = 0
    for q, wd in enumerate(widths):
        s = d.seg[q]
        s.M, s.N, s.K, s.ldc, s.Mvalid, s.C = M, N, wd, N, M, y.data_ptr()
        if wd:
            s.A, s.B, s.lda, s.ldb = xs[q].data_ptr(), w.data_ptr() + 4 * off, wd, K
        off += wd
    assert P.gemm_kernel_name(d) == "gemm_kslice_kernel", "the case must be sized for the single-pass kernel"
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()
    live = [q for q, wd in enumerate(widths) if wd]
    x64 = torch.cat([xs[q].double() for q in live], 1)
    cols = torch.cat([torch.arange(sum(widths[:q]), sum(widths[:q]) + widths[q], device="cuda") for q in live])
    pre = x64 @ w.double()[:, cols].t() + (b.double() if b is not None else 0.0)
    ref = {L.ACT_NONE: pre, L.ACT_RELU: torch.relu(pre), L.ACT_SILU: pre * torch.sigmoid(pre)}[act]
    if beta:
        ref = ref + y0.double()
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 2e-6 * max(scale, 1.0) * max(1.0, K ** 0.5 / 16)  # fp32 accumulation over K
    if z is not None:
        assert float((z.double() - pre).abs().max()) <= 2e-6 * max(float(pre.abs().max()), 1.0) * max(1.0, K ** 0.5 / 16)
    # the summation order is a fixed function of the segment list: a second launch reproduces every bit
    first = y.clone()
    y.copy_(y0)
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()
    assert torch.equal(y, first)

