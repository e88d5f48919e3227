"""The throughput-regime GEMM (csrc/gemm_fast.hip: 128x128x32 tiles, buffer-load staging) through the C-ABI against fp64 products:
the three operand bindings of the nn.Linear family (y = x W^T — modules.py:171,489,515,584; dx = dy W; dW = dy^T x), K-segmented
inputs, batched problems, the bias gradient's ones-column, row masks, accumulation, the full epilogue of the gated product
(bias, sigmoid, multiplier segments, saved pre-activations, prefix mask), split-K, and the balanced schedule
(NASREC_SPLITK_BALANCED) for tile counts below, between and above the 512-workgroup rounds — which must also be deterministic."""
import ctypes as C

import pytest
import torch

from nasrec_amd import _lib as L
from nasrec_amd import plan as P

pytestmark = pytest.mark.gpu

SCHEDULES = [1, 2, L.SPLITK_BALANCED]


@pytest.fixture(scope="module")
def lib():
    return L.load()


@pytest.fixture(scope="module")
def sk_ws():
    return torch.empty(L.SK_WORKSPACE_FLOATS, dtype=torch.float32, device="cuda")


def _rand(*shape, scale=1.0):
    return (torch.randn(*shape, device="cuda") * scale).contiguous()


def _desc(am, bm, segs, zmode, splitk, sk_ws, keep, **kw):
    d = L.GemmDesc()
    d.kind = L.OP_GEMM
    d.amode, d.bmode, d.cmode, d.nseg, d.zmode = am, bm, L.CM_PLAIN, len(segs), zmode
    d.dims_in_use = kw.get("dims", -1)
    d.act = kw.get("act", 0)
    d.beta = kw.get("beta", 0)
    for k in ("bias", "save_z", "save_act", "rowsum_out"):
        if kw.get(k) is not None:
            setattr(d, k, kw[k].data_ptr())
    for q, (ptr, off, width, ld) in enumerate(kw.get("mul", [])):
        d.mul_ptr[q], d.mul_off[q], d.mul_width[q], d.mul_ld[q] = ptr, off, width, ld
    d.mul_nseg = len(kw.get("mul", []))
    for q, sd in enumerate(segs):
        for k, v in sd.items():
            setattr(d.seg[q], k, v)
        if "Mvalid" not in sd:
            d.seg[q].Mvalid = sd["M"]
    d.splitk = splitk
    if splitk == L.SPLITK_BALANCED:
        d.workspace = sk_ws.data_ptr()
    elif splitk > 1:
        nprob = len(segs) if zmode else 1
        ws = torch.empty(splitk * max(s["M"] for s in segs) * max(s["N"] for s in segs) * nprob, device="cuda")
        keep.append(ws)
        d.workspace = ws.data_ptr()
    assert P.gemm_kernel_name(d) == "gemm_fast_kernel", "the case must be sized for the throughput kernel"
    return d


def _launch(lib, d):
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()


def _close(got, want, tol=2e-5):
    scale = max(1.0, float(want.abs().max()))
    err = float((got.double() - want).abs().max())
    assert err <= tol * scale, "max err %.3e (scale %.3e)" % (err, scale)


@pytest.mark.parametrize("splitk", SCHEDULES)
def test_forward_product_k_segments_and_the_gated_epilogue(lib, sk_ws, splitk):
    torch.manual_seed(1)
    M, N, Ks = 2048, 1035, [13, 1024, 300]   # 16 x 9 = 144 tiles; the last tile column is 11 wide, K tails of 13 and 12
    xs = [_rand(M, k + 3)[:, :k] for k in Ks]            # row strides that are not multiples of 4 floats
    Ws = [_rand(N, k, scale=0.05) for k in Ks]
    bias, R1, R2 = _rand(N), _rand(M, 700), _rand(M, 200)
    out, z, a = torch.full((M, N + 5), 7.0, device="cuda"), torch.zeros(M, N + 5, device="cuda"), torch.zeros(M, N + 5, device="cuda")
    keep = []
    segs = [dict(A=x.data_ptr(), B=W.data_ptr(), C=out.data_ptr(), M=M, N=N, K=k, lda=x.stride(0), ldb=k, ldc=N + 5) for x, W, k in zip(xs, Ws, Ks)]
    dims = 1000
    d = _desc(L.AM_KC, L.AM_KC, segs, 0, splitk, sk_ws, keep, bias=bias, act=L.ACT_SIGMOID, dims=dims, save_z=z, save_act=a,
              mul=[(R1.data_ptr(), 0, 700, 700), (R2.data_ptr(), 800, 200, 200)])
    _launch(lib, d)
    zz = sum(x.double() @ W.double().t() for x, W in zip(xs, Ws)) + bias.double()
    R = torch.zeros(M, N, dtype=torch.float64, device="cuda")
    R[:, :700] = R1.double()
    R[:, 800:1000] = R2.double()
    want = torch.sigmoid(zz) * R
    want[:, dims:] = 0
    _close(z[:, :N], zz)
    _close(a[:, :N], torch.sigmoid(zz))
    _close(out[:, :N], want)
    assert bool((out[:, N:] == 7.0).all()), "columns beyond N were written"


@pytest.mark.parametrize("splitk", SCHEDULES)
def test_input_gradient_batch(lib, sk_ws, splitk):
    torch.manual_seed(2)
    B, K = 2048, 512
    shapes = [(1035, True), (1024, False)]     # (in features, accumulate)
    keep, segs, checks = [], [], []
    for nin, acc in shapes:
        dy, W, dx = _rand(B, K), _rand(K, nin, scale=0.05), _rand(B, nin)
        dx0 = dx.clone()
        segs.append(dict(A=dy.data_ptr(), B=W.data_ptr(), C=dx.data_ptr(), M=B, N=nin, K=K, lda=K, ldb=nin, ldc=nin, accumulate=int(acc)))
        keep.append((dy, W))
        checks.append((dx, (dx0.double() if acc else 0) + dy.double() @ W.double()))
    _launch(lib, _desc(L.AM_KC, L.AM_RC, segs, 1, splitk, sk_ws, keep))
    for got, want in checks:
        _close(got, want)


@pytest.mark.parametrize("splitk", SCHEDULES)
def test_weight_gradient_batch_with_bias_column_row_mask_and_accumulation(lib, sk_ws, splitk):
    torch.manual_seed(3)
    B = 2048
    shapes = [(1024, 1034, 1024, 0, 1), (512, 1024, 400, 1, 0), (128, 1024, 128, 0, 0), (1024, 1024, 1024, 1, 1)]  # nout, nin, Mvalid, acc, ones
    keep, segs, checks = [], [], []
    for nout, nin, mv, acc, ones in shapes:
        dy, x, dW, db = _rand(B, nout, scale=0.3), _rand(B, nin), _rand(nout, nin), torch.zeros(nout, device="cuda")
        dW0 = dW.clone()
        segs.append(dict(A=dy.data_ptr(), B=x.data_ptr(), C=dW.data_ptr(), M=nout, N=nin + ones, K=B, lda=nout, ldb=nin, ldc=nin, Mvalid=mv,
                         accumulate=acc, ones_col=ones, rowsum=db.data_ptr() if ones else None))
        keep.append((dy, x))
        dz = dy.double().clone()
        dz[:, mv:] = 0
        checks.append((dW, (dW0.double() if acc else 0) + dz.t() @ x.double(), db if ones else None, dz.sum(0)))
    _launch(lib, _desc(L.AM_RC, L.AM_RC, segs, 1, splitk, sk_ws, keep))
    for dW, want_w, db, want_b in checks:
        _close(dW, want_w, tol=5e-5)
        if db is not None:
            _close(db, want_b, tol=5e-5)


@pytest.mark.parametrize("M,N,K", [(4096, 4224, 100), (4096, 2200, 200), (2048, 1100, 4000)])
def test_balanced_schedule_across_round_boundaries_is_exact_and_deterministic(lib, sk_ws, M, N, K):
    """1056 tiles (544 shared + 512 whole, shares of 4.25 k-tiles cross tile boundaries), 576 tiles (all shared), 144 tiles of 125
    k-tiles (every tile split over 3-4 workgroups); same inputs -> same bits, launch after launch"""
    torch.manual_seed(4)
    x, W = _rand(M, K), _rand(N, K, scale=0.05)
    keep, outs = [], []
    want = x.double() @ W.double().t()
    for splitk in (L.SPLITK_BALANCED, 1, L.SPLITK_BALANCED):
        y = torch.full((M, N), float("nan"), device="cuda")
        sk_ws.fill_(float("nan"))
        d = _desc(L.AM_KC, L.AM_KC, [dict(A=x.data_ptr(), B=W.data_ptr(), C=y.data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N)], 0, splitk, sk_ws, keep)
        for _ in range(2):  # back to back: the second launch reuses the workspace
            L.check(lib.nasrec_launch(None, C.addressof(d)))
        torch.cuda.synchronize()
        _close(y, want)
        outs.append(y)
    assert torch.equal(outs[0], outs[2])


def test_balanced_schedule_is_refused_where_it_does_not_apply(lib, sk_ws):
    x, W, y = _rand(64, 64), _rand(64, 64), _rand(64, 64)
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use = L.OP_GEMM, L.AM_KC, L.AM_KC, L.CM_PLAIN, 1, 0, -1
    for k, v in dict(A=x.data_ptr(), B=W.data_ptr(), C=y.data_ptr(), M=64, N=64, K=64, lda=64, ldb=64, ldc=64, Mvalid=64).items():
        setattr(d.seg[0], k, v)
    d.splitk, d.workspace = L.SPLITK_BALANCED, sk_ws.data_ptr()
    with pytest.raises(L.EngineError):
        L.check(lib.nasrec_launch(None, C.addressof(d)))
    # a batch with unequal K
    a, b = _rand(2048, 256), _rand(2048, 512)
    Wa, Wb = _rand(1024, 256), _rand(1024, 512)
    ya, yb = _rand(2048, 1024), _rand(2048, 1024)
    segs = [dict(A=a.data_ptr(), B=Wa.data_ptr(), C=ya.data_ptr(), M=2048, N=1024, K=256, lda=256, ldb=256, ldc=1024),
            dict(A=b.data_ptr(), B=Wb.data_ptr(), C=yb.data_ptr(), M=2048, N=1024, K=512, lda=512, ldb=512, ldc=1024)]
    d2 = _desc(L.AM_KC, L.AM_KC, segs, 1, L.SPLITK_BALANCED, sk_ws, [])
    with pytest.raises(L.EngineError):
        L.check(lib.nasrec_launch(None, C.addressof(d2)))
    assert not P._balanced_schedule_pays([dict(A=1, M=2048, N=1024, K=256), dict(A=1, M=2048, N=1024, K=512)], 1)
    assert P._balanced_schedule_pays([dict(A=1, M=1024, N=1035, K=4096)] * 2 + [dict(A=1, M=1024, N=1024, K=4096)] * 6, 1)   # 528 tiles
    assert not P._balanced_schedule_pays([dict(A=1, M=1024, N=1024, K=4096)] * 8, 1)                                        # 512 tiles


# ---------------------------------------------------------------------------------------------------------------------------------
# token-axis Linear at large batch (csrc/token_linear.hip): weights in LDS, a wavefront per sample
# ---------------------------------------------------------------------------------------------------------------------------------
def _tok_desc(am, segs, zmode, **kw):
    d = L.GemmDesc()
    d.kind = L.OP_GEMM
    d.amode, d.bmode, d.cmode, d.nseg, d.zmode = am, L.AM_TOKR, L.CM_TOKJ, len(segs), zmode
    d.dims_in_use = kw.get("dims", -1)
    d.act = kw.get("act", 0)
    d.beta = kw.get("beta", 0)
    d.bias_on_rows, d.mask_on_rows = 1, 1
    d.splitk = 1
    for k in ("bias", "save_z"):
        if kw.get(k) is not None:
            setattr(d, k, kw[k].data_ptr())
    for q, sd in enumerate(segs):
        for k, v in sd.items():
            setattr(d.seg[q], k, v)
        d.seg[q].Mvalid = sd["M"]
    assert P.gemm_kernel_name(d) == "token_linear_kernel"
    return d


@pytest.mark.parametrize("nout,act,dims,beta", [(64, L.ACT_NONE, -1, 0), (45, L.ACT_RELU, 30, 0), (80, L.ACT_SILU, -1, 1), (7, L.ACT_NONE, -1, 0), (33, L.ACT_NONE, 33, 0)])
def test_token_linear_forward_over_input_segments(lib, nout, act, dims, beta):
    """out[b, n', e] = act(sum_n W[n', n] x[b, n, e] + bias[n']) with a prefix mask over n' — modules.py:222-234; inputs are token
    sub-ranges of three slabs with different batch strides, K = 26 + 72 + 9 (a ragged last k-step in every segment)"""
    torch.manual_seed(5)
    B, Ks = 1500, ([26, 72, 9] if nout != 64 else [72, 72, 72, 72, 21])   # nout 64: 99 KB of staged weights (> the default LDS limit)
    Ntot = sum(Ks)
    W, bias = _rand(nout, Ntot, scale=0.2), _rand(nout)
    slabs = [_rand(B, k + 5, 16) for k in Ks]                   # the segment is rows 2 .. 2+k of a larger slab
    out = _rand(B, nout + 3, 16)
    out0 = out.clone()
    z = torch.zeros_like(out)
    segs, koff = [], 0
    for slab, k in zip(slabs, Ks):
        segs.append(dict(A=W.data_ptr() + 4 * koff, B=slab.data_ptr() + 4 * 2 * 16, C=out.data_ptr() + 4 * 16, M=nout, N=B * 16, K=k, lda=Ntot,
                         ldb=slab.stride(0), ldc=out.stride(0)))
        koff += k
    d = _tok_desc(L.AM_KC, segs, 0, bias=bias, act=act, dims=dims, beta=beta, save_z=z if act == L.ACT_SILU else None)
    if act == L.ACT_SILU:
        d.save_z = z.data_ptr() + 4 * 16
    _launch(lib, d)
    x = torch.cat([slab[:, 2:2 + k] for slab, k in zip(slabs, Ks)], 1).double()
    zz = torch.einsum("on,bne->boe", W.double(), x) + bias.double()[None, :, None]
    want = {L.ACT_NONE: zz, L.ACT_RELU: zz.clamp_min(0), L.ACT_SILU: zz * torch.sigmoid(zz)}[act].clone()
    if dims >= 0:
        want[:, dims:] = 0
    if beta:
        want = want + out0[:, 1:1 + nout].double()
    _close(out[:, 1:1 + nout], want)
    assert torch.equal(out[:, 0], out0[:, 0]) and torch.equal(out[:, 1 + nout:], out0[:, 1 + nout:]), "rows outside the target were written"
    if act == L.ACT_SILU:
        _close(z[:, 1:1 + nout], zz)


def test_token_linear_input_gradient_batch(lib):
    """dx_s[b, n, e] = sum_n' W[n', koff_s + n] dz[b, n', e] for every input segment s (one launch, independent problems), with the
    prefix mask as a shorter K and accumulation into a gradient that already holds a contribution"""
    torch.manual_seed(6)
    B, nout, kd, Ks = 2048, 64, 50, [72, 72, 66]
    Ntot = sum(Ks)
    W, dz = _rand(nout, Ntot, scale=0.2), _rand(B, nout, 16)
    segs, checks, koff = [], [], 0
    for q, k in enumerate(Ks):
        dx = _rand(B, k + 2, 16)
        dx0 = dx.clone()
        acc = q == 1
        segs.append(dict(A=W.data_ptr() + 4 * koff, B=dz.data_ptr(), C=dx.data_ptr(), M=k, N=B * 16, K=kd, lda=Ntot, ldb=dz.stride(0),
                         ldc=dx.stride(0), accumulate=int(acc)))
        want = torch.einsum("on,boe->bne", W[:kd, koff:koff + k].double(), dz[:, :kd].double())
        checks.append((dx, (dx0[:, :k].double() if acc else 0) + want, dx0, k))
        koff += k
    _launch(lib, _tok_desc(L.AM_RC, segs, 1))
    for dx, want, dx0, k in checks:
        _close(dx[:, :k], want)
        assert torch.equal(dx[:, k:], dx0[:, k:])


@pytest.mark.parametrize("S", [8, 37])
def test_token_weight_gradient_batch_at_large_batch(lib, S):
    """dW[n', n] = sum_{b,e} dz[b, n', e] x[b, n, e] for a batch of input segments of one token-axis Linear (modules.py:222-234 backward):
    bias gradient as the ones-column of the first problem, a prefix mask over n' (Mvalid), accumulation over an earlier product,
    operands that are token sub-ranges of larger slabs; S workgroups per problem -> S slabs -> the fixed-order second pass"""
    torch.manual_seed(7)
    B, nout, kd = 1280, 45, 40
    dz = _rand(B, nout + 2, 16, scale=0.3)
    widths = [72, 26, 9, 72]
    Ntot = sum(widths)
    dW, db = _rand(nout, Ntot), torch.zeros(nout, device="cuda")
    dW0 = dW.clone()
    segs, checks, koff = [], [], 0
    for q, w in enumerate(widths):
        slab = _rand(B, w + 4, 16)
        ones, acc = int(q == 0), int(q == 3)
        segs.append(dict(A=dz.data_ptr() + 4 * 16, B=slab.data_ptr() + 4 * 3 * 16, C=dW.data_ptr() + 4 * koff, M=nout, N=w + ones, K=B * 16,
                         lda=dz.stride(0), ldb=slab.stride(0), ldc=Ntot, Mvalid=kd, accumulate=acc, ones_col=ones, rowsum=db.data_ptr() if ones else None))
        g = dz[:, 1:1 + nout].double().clone()
        g[:, kd:] = 0
        want = torch.einsum("boe,bne->on", g, slab[:, 3:3 + w].double())
        checks.append((koff, w, (dW0[:, koff:koff + w].double() if acc else 0) + want, slab))
        koff += w
    d = L.GemmDesc()
    d.kind = L.OP_GEMM
    d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use = L.AM_TOKK, L.AM_TOKK, L.CM_PLAIN, len(segs), 1, -1
    for q, sd in enumerate(segs):
        for k, v in sd.items():
            setattr(d.seg[q], k, v)
    ws = torch.full((S * nout * 73 * len(segs),), float("nan"), device="cuda")
    d.splitk, d.workspace = S, ws.data_ptr()
    _launch(lib, d)
    for koff, w, want, _ in checks:
        _close(dW[:, koff:koff + w], want, tol=5e-5)
    g = dz[:, 1:1 + nout].double().clone()
    g[:, kd:] = 0
    _close(db, g.sum((0, 2)), tol=5e-5)
