"""Worklist items against the stand-alone kernels on the SAME descriptor, bit for bit (csrc/worklist_body.h): the per-sample token-axis
bodies (wl_token_fwd: W x, modules.py:222-234; wl_token_dx: W^T dy per input segment) on ragged shapes — rows of W that are not a
multiple of 16, token segments that end inside a 16-token step, a handful of samples — and the general tile with its 16-byte staging
(k-contiguous and row-contiguous operands, mask operands, split-K halves are covered by the network-level bit-identity tests)."""
import ctypes as C

import pytest
import torch

from nasrec_amd import _lib as L
from nasrec_amd import schedule as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return L.load()


def _as_item(d):
    n = S.Node(d)
    b = S.item_bytes(n)
    assert b is not None
    one = L.WorklistDesc()
    one.kind, one.n = L.OP_WORKLIST, 1
    it = one.item[0]
    it.kind, it.part, it.off = d.kind, L.WL_WHOLE, 0
    C.memmove(C.addressof(one) + L.WorklistDesc.blob.offset, b, len(b))
    return one


def _run(lib, d):
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()


_KEEP = []


def _resident(lib, w):
    """the same worklist launch with its descriptor uploaded once (ABI 17: nasrec_worklist_prepare -> NASREC_OP_WORKLIST_DEV)"""
    buf = torch.empty(C.sizeof(L.WorklistDesc), dtype=torch.uint8, device="cuda")
    dv = L.WorklistDevDesc()
    torch.cuda.synchronize()
    L.check(lib.nasrec_worklist_prepare(C.addressof(w), buf.data_ptr(), C.addressof(dv)))
    assert dv.kind == L.OP_WORKLIST_DEV and dv.total_blocks > 0 and dv.dev == buf.data_ptr()
    _KEEP.append(buf)
    return dv


@pytest.mark.parametrize("B,M,toks,bias,act", [(256, 39, [26, 72, 64], True, L.ACT_NONE), (3, 8, [26], False, L.ACT_RELU), (17, 64, [5, 16, 33], True, L.ACT_NONE),
                                               (256, 48, [26, 72, 64, 32], False, L.ACT_NONE), (64, 16, [72, 56], True, L.ACT_RELU)])
def test_token_forward_item_is_bit_identical(lib, B, M, toks, bias, act):
    g = torch.Generator(device="cuda").manual_seed(B + M)
    K = sum(toks)
    w = torch.randn(M, K, device="cuda", generator=g)
    xs = [torch.randn(B, t, 16, device="cuda", generator=g) for t in toks]
    bvec = torch.randn(M, device="cuda", generator=g) if bias else None
    outs = []
    for as_item in (False, True):
        y = torch.zeros(B, M, 16, device="cuda")
        d = L.GemmDesc()
        d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use, d.splitk = L.OP_GEMM, L.AM_KC, L.AM_TOKR, L.CM_TOKJ, len(toks), 0, -1, 1
        d.act = act
        if bvec is not None:
            d.bias, d.bias_on_rows = bvec.data_ptr(), 1
        off = 0
        for q, t in enumerate(toks):
            s = d.seg[q]
            s.A, s.B, s.C, s.M, s.N, s.K, s.lda, s.ldb, s.ldc, s.Mvalid = w.data_ptr() + 4 * off, xs[q].data_ptr(), y.data_ptr(), M, B * 16, t, K, t * 16, M * 16, M
            off += t
        _run(lib, _as_item(d) if as_item else d)
        outs.append(y)
    ref = torch.einsum("mk,bke->bme", w.double(), torch.cat(xs, 1).double()) + (bvec.double().view(1, M, 1) if bvec is not None else 0.0)
    if act == L.ACT_RELU:
        ref = torch.relu(ref)
    assert float((outs[0].double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("mask", [False, True])
@pytest.mark.parametrize("B,rows_out,toks,acc", [(256, 32, [26, 72, 64], False), (5, 48, [72, 32], True), (256, 64, [26], False), (33, 16, [27, 5], True)])
def test_token_input_gradient_item_is_bit_identical(lib, B, rows_out, toks, acc, mask):
    """stand-alone kernel, worklist item (descriptor by value) and worklist item with the descriptor resident in device memory; with
    `mask` the gradient counts as 0 where the Linear's saved activation is <= 0 (the fused ReLU backward: Baux), which the
    wavefront-per-tile body applies as it loads the operand (round 6; the general tile before)"""
    g = torch.Generator(device="cuda").manual_seed(B * 3 + rows_out)
    K = sum(toks)
    w = torch.randn(rows_out, K, device="cuda", generator=g)       # the Linear's weight [out rows, tokens]
    dy = torch.randn(B, rows_out, 16, device="cuda", generator=g)  # gradient of its output
    act = torch.randn(B, rows_out, 16, device="cuda", generator=g)  # its saved activation (the mask)
    init = [torch.randn(B, t, 16, device="cuda", generator=g) for t in toks]
    outs = []
    for as_item in (False, True, "resident"):
        dx = [t.clone() for t in init]
        d = L.GemmDesc()
        d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use, d.splitk = L.OP_GEMM, L.AM_RC, L.AM_TOKR, L.CM_TOKJ, len(toks), 1, -1, 1
        off = 0
        for q, t in enumerate(toks):
            s = d.seg[q]
            s.A, s.B, s.C, s.M, s.N, s.K, s.lda, s.ldb, s.ldc, s.Mvalid = w.data_ptr() + 4 * off, dy.data_ptr(), dx[q].data_ptr(), t, B * 16, rows_out, K, rows_out * 16, t * 16, t
            s.accumulate = 1 if acc else 0
            if mask:
                s.Baux = act.data_ptr()
            off += t
        _run(lib, d if not as_item else (_resident(lib, _as_item(d)) if as_item == "resident" else _as_item(d)))
        outs.append(dx)
    off = 0
    dz = dy.double() * ((act > 0).double() if mask else 1.0)
    for q, t in enumerate(toks):
        ref = torch.einsum("ik,bie->bke", w[:, off:off + t].double(), dz) + (init[q].double() if acc else 0.0)
        assert float((outs[0][q].double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
        assert torch.equal(outs[0][q], outs[1][q])
        assert torch.equal(outs[0][q], outs[2][q])
        off += t
