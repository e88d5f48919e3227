"""Per-kernel parity tests of the HIP engine through the C-ABI (run with `-m gpu` on an MI355X).
Each test compares one descriptor launch against plain torch fp64 math of the same operator on the same seeded inputs.
Integer/index work must be bit-exact; fp32 work within 2e-5 relative to the operand scale (stated per test)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from nasrec_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return L.load()


def dev(t):
    return t.to("cuda").contiguous()


def launch(lib, d):
    L.check(lib.nasrec_launch(None, C.addressof(d)))
    torch.cuda.synchronize()


def close(a, b, tol=2e-5):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, "max err %.3e (scale %.3e)" % (err, scale)


_KEEP = []


def gemm_desc(am, bm, cm, segs, zmode, **kw):
    d = L.GemmDesc()
    d.kind = L.OP_GEMM
    d.amode, d.bmode, d.cmode, d.nseg, d.zmode = am, bm, cm, len(segs), zmode
    d.dims_in_use = kw.get("dims", -1)
    d.act = kw.get("act", 0)
    d.bias = kw.get("bias", None)
    d.bias_on_rows = kw.get("bias_on_rows", 0)
    d.mask_on_rows = kw.get("mask_on_rows", 0)
    d.beta = kw.get("beta", 0)
    d.splitk = kw.get("splitk", 1)
    d.workspace = kw.get("workspace", None)
    d.rowsum_out = kw.get("rowsum_out", None)
    for q, sd in enumerate(segs):
        for k, v in sd.items():
            setattr(d.seg[q], k, v)
        if "Mvalid" not in sd:
            d.seg[q].Mvalid = sd["M"]
    return d


@pytest.mark.parametrize("splitk", [1, 3])
@pytest.mark.parametrize("act,dims", [(L.ACT_NONE, -1), (L.ACT_RELU, 40)])
def test_gemm_dense_forward_segments(lib, splitk, act, dims):
    """y = act(cat(x0, 0, x2) Wᵀ + b) * mask — ragged widths (13, 7, 781) exercise the tile edges."""
    torch.manual_seed(0)
    B, N = 70, 77
    w0, w1, w2 = 13, 7, 781
    x0 = torch.randn(B, w0)
    x2s = torch.randn(B, w2 + 5)  # x2 has a row stride != width
    x2 = x2s[:, :w2]
    W, b = torch.randn(N, w0 + w1 + w2) * 0.1, torch.randn(N)
    y0 = torch.randn(B, N)
    gx0, gx2s, gW, gb, gy = dev(x0), dev(x2s), dev(W), dev(b), dev(y0)
    K = w0 + w1 + w2
    segs = [dict(A=gx0.data_ptr(), B=gW.data_ptr(), C=gy.data_ptr(), M=B, N=N, K=w0, lda=w0, ldb=K, ldc=N),
            dict(A=None, B=gW.data_ptr() + 4 * w0, C=gy.data_ptr(), M=B, N=N, K=w1, lda=1, ldb=K, ldc=N),
            dict(A=gx2s.data_ptr(), B=gW.data_ptr() + 4 * (w0 + w1), C=gy.data_ptr(), M=B, N=N, K=w2, lda=w2 + 5, ldb=K, ldc=N)]
    ws = dev(torch.zeros(splitk * B * N))
    d = gemm_desc(L.AM_KC, L.AM_KC, L.CM_PLAIN, segs, 0, act=act, dims=dims, bias=gb.data_ptr(), beta=1, splitk=splitk,
                  workspace=ws.data_ptr())
    launch(lib, d)
    xcat = torch.cat([x0, torch.zeros(B, w1), x2], 1).double()
    ref = xcat @ W.double().t() + b.double()
    if act == L.ACT_RELU:
        ref = ref.clamp_min(0)
    if dims >= 0:
        ref[:, dims:] = 0
    ref = ref + y0.double()
    close(gy, ref)


@pytest.mark.parametrize("splitk", [1, 3])
def test_gemm_zmode_batch_of_unequal_weight_gradients_every_tile_configuration(lib, splitk):
    """A parked-weight-gradient style launch: independent dW = (dy ⊙ [y>0])ᵀ x products of very different sizes in ONE zmode
    launch (ReLU-mask operand, row prefix mask, accumulate, per-problem bias gradient as a virtual ones-column), sized so
    that the launcher picks, in turn, the skinny tiles, the 32x32 batch tiles of the latency regime and the 64x64 throughput
    tiles (live 64x64 workgroups < 128, 128..1023, >= 1024)."""
    torch.manual_seed(21)
    for scale, B in ((1, 96), (4, 256), (12, 160)):
        shapes = [(192 * scale, 195 * scale, 150 * scale), (64, 37, 64), (13, 14, 9), (200, 129, 111)]  # (nout, nin, dims)
        segs, checks, keep = [], [], []
        for q, (nout, nin, dims) in enumerate(shapes):
            dy, y, x = torch.randn(B, nout) * 0.3, torch.randn(B, nout), torch.randn(B, nin)
            dw0 = torch.randn(nout, nin)
            g = [dev(t) for t in (dy, y, x, dw0.clone(), torch.zeros(nout))]
            keep.append(g)
            acc = q % 2
            segs.append(dict(A=g[0].data_ptr(), Aaux=g[1].data_ptr(), B=g[2].data_ptr(), C=g[3].data_ptr(), M=nout, N=nin + 1, K=B, lda=nout,
                             ldb=nin, ldc=nin, Mvalid=dims, accumulate=acc, ones_col=1, rowsum=g[4].data_ptr()))
            dz = (dy * (y > 0)).double()
            dz[:, dims:] = 0
            checks.append((g[3], (dw0.double() if acc else 0) + dz.t() @ x.double(), g[4], dz.sum(0)))
        Mm, Nm = max(s["M"] for s in segs), max(s["N"] for s in segs)
        ws = dev(torch.zeros(splitk * Mm * Nm * len(segs)))
        launch(lib, gemm_desc(L.AM_RC, L.AM_RC, L.CM_PLAIN, segs, 1, splitk=splitk, workspace=ws.data_ptr()))
        for dw, want_w, db, want_b in checks:
            close(dw, want_w)
            close(db, want_b)


@pytest.mark.parametrize("splitk", [4, 16])
def test_gemm_token_weight_gradient_batch_in_the_latency_regime(lib, splitk):
    """token-axis dW[p, n] = sum_{b,e} (dy ⊙ [y>0])[b,p,e] x[b,n,e] for several Transformer / Linear3D projections in one zmode
    launch at batch 256 (K = 4096): the 32x32 batch tiles with split-K"""
    torch.manual_seed(22)
    B = 256
    shapes = [(48, 73, 40), (64, 27, 64), (16, 104, 9), (39, 162, 39), (8, 26, 8), (32, 64, 20)]  # (tokens out, tokens in, dims)
    segs, checks, keep = [], [], []
    for q, (Np, n, dims) in enumerate(shapes):
        dy, y, x = torch.randn(B, Np, 16) * 0.3, torch.randn(B, Np, 16), torch.randn(B, n, 16)
        g = [dev(t) for t in (dy, y, x, torch.zeros(Np, n), torch.zeros(Np))]
        keep.append(g)
        segs.append(dict(A=g[0].data_ptr(), Aaux=g[1].data_ptr(), B=g[2].data_ptr(), C=g[3].data_ptr(), M=Np, N=n + 1, K=B * 16, lda=Np * 16,
                         ldb=n * 16, ldc=n, Mvalid=dims, ones_col=1, rowsum=g[4].data_ptr()))
        dz = (dy * (y > 0)).double()
        dz[:, dims:] = 0
        checks.append((g[3], torch.einsum("bpe,bne->pn", dz, x.double()), g[4], dz.sum((0, 2))))
    Mm, Nm = max(s["M"] for s in segs), max(s["N"] for s in segs)
    ws = dev(torch.zeros(splitk * Mm * Nm * len(segs)))
    launch(lib, gemm_desc(L.AM_TOKK, L.AM_TOKK, L.CM_PLAIN, segs, 1, splitk=splitk, workspace=ws.data_ptr()))
    for dw, want_w, db, want_b in checks:
        close(dw, want_w)
        close(db, want_b)


def test_gemm_dense_backward_products(lib):
    """dx_seg (+)= (dy ⊙ [y>0])[:, :dims] W[:dims, seg];  dW[:, seg] = (dy ⊙ [y>0])ᵀ x_seg with rows >= dims zero."""
    torch.manual_seed(1)
    B, N, dims = 66, 50, 33
    w0, w1 = 21, 100
    K = w0 + w1
    x0, x1, W = torch.randn(B, w0), torch.randn(B, w1), torch.randn(N, K) * 0.1
    y = torch.randn(B, N)
    y[:, dims:] = 0
    dy = torch.randn(B, N)
    dx0_init = torch.randn(B, w0)
    g = {k: dev(v) for k, v in dict(x0=x0, x1=x1, W=W, y=y, dy=dy, dx0=dx0_init.clone(), dx1=torch.zeros(B, w1), dW=torch.full((N, K), 7.0)).items()}
    zs = [dict(A=g["dy"].data_ptr(), Aaux=g["y"].data_ptr(), B=g["W"].data_ptr(), C=g["dx0"].data_ptr(), M=B, N=w0, K=dims, lda=N, ldb=K, ldc=w0, accumulate=1),
          dict(A=g["dy"].data_ptr(), Aaux=g["y"].data_ptr(), B=g["W"].data_ptr() + 4 * w0, C=g["dx1"].data_ptr(), M=B, N=w1, K=dims, lda=N, ldb=K, ldc=w1, accumulate=0)]
    launch(lib, gemm_desc(L.AM_KC, L.AM_RC, L.CM_PLAIN, zs, 1))
    dz = (dy * (y > 0)).double()
    dz[:, dims:] = 0
    close(g["dx0"], dx0_init.double() + dz @ W.double()[:, :w0])
    close(g["dx1"], dz @ W.double()[:, w0:])
    ws_ = [dict(A=g["dy"].data_ptr(), Aaux=g["y"].data_ptr(), B=g["x0"].data_ptr(), C=g["dW"].data_ptr(), M=N, N=w0, K=B, lda=N, ldb=w0, ldc=K, Mvalid=dims),
           dict(A=g["dy"].data_ptr(), Aaux=g["y"].data_ptr(), B=g["x1"].data_ptr(), C=g["dW"].data_ptr() + 4 * w0, M=N, N=w1, K=B, lda=N, ldb=w1, ldc=K, Mvalid=dims)]
    launch(lib, gemm_desc(L.AM_RC, L.AM_RC, L.CM_PLAIN, ws_, 1))
    close(g["dW"], dz.t() @ torch.cat([x0, x1], 1).double())
    # bias gradient
    db = dev(torch.zeros(N))
    r = L.RowsumDesc()
    r.kind, r.mode, r.R, r.K, r.ld, r.rvalid = L.OP_ROWSUM, L.AM_RC, N, B, N, dims
    r.p, r.aux, r.out = g["dy"].data_ptr(), g["y"].data_ptr(), db.data_ptr()
    launch(lib, r)
    close(db, dz.sum(0))


@pytest.mark.parametrize("B", [8, 256])
@pytest.mark.parametrize("specs", [[(48, 0, False), (45, 0, False)], [(48, 192, True), (45, 128, False)], [(48, 0, True), (45, 0, True)],
                                   [(32, 5, False), (39, 7, True), (64, 0, False), (8, 0, True)]])
def test_gemm_token_input_gradients_concatenated_along_k(lib, B, specs):
    """dx[b, n, e] = sum over several token-axis Linears l of sum_n' W_l[n', n] dz_l[b, n', e] as ONE launch: binding RC / TOKR / TOKJ with
    k-segments that differ in K, in the row stride of their weights and in having a ReLU-mask operand (the parked gradients into the raw
    embedding tokens, plan._flush_raw_dx).  (k-segments that disagree on the mask operand once went through the staged ring, which keeps
    that flag per launch segment in flight: wrong results.)"""
    torch.manual_seed(30)
    M, N = 10, B * 16
    out = dev(torch.zeros(B, M, 16))
    d = L.GemmDesc()
    d.kind = L.OP_GEMM
    d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use, d.splitk = L.AM_RC, L.AM_TOKR, L.CM_TOKJ, len(specs), 0, -1, 1
    want = torch.zeros(B, M, 16, dtype=torch.float64)
    keep = []
    for q, (K, extra, aux) in enumerate(specs):
        lda = M + extra
        W, dz, y = torch.randn(K, lda), torch.randn(B, K + 3, 16), torch.randn(B, K + 3, 16)
        gW, gdz, gy = dev(W), dev(dz), dev(y)
        keep += [gW, gdz, gy]
        s = d.seg[q]
        s.A, s.B, s.C = gW.data_ptr(), gdz.data_ptr(), out.data_ptr()
        s.M, s.N, s.K, s.lda, s.ldb, s.ldc, s.Mvalid = M, N, K, lda, (K + 3) * 16, M * 16, M
        g = dz[:, :K].double()
        if aux:
            s.Baux = gy.data_ptr()
            g = g * (y[:, :K] > 0).double()
        want += torch.einsum("ki,bke->bie", W[:, :M].double(), g)
    launch(lib, d)
    close(out, want, 2e-5)


@pytest.mark.parametrize("splitk", [1, 4])
def test_gemm_token_axis_all_products(lib, splitk):
    """token-axis Linear over [B,N,16] with two token segments living inside larger slabs, + both gradients."""
    torch.manual_seed(2)
    B, n0, n1, Np, dims = 9, 26, 45, 40, 24
    slab0, slab1 = torch.randn(B, n0 + 3, 16), torch.randn(B, n1, 16)
    x0, x1 = slab0[:, 3:, :], slab1
    Nt = n0 + n1
    W, bias = torch.randn(Np, Nt) * 0.2, torch.randn(Np)
    g0, g1, gW, gb = dev(slab0), dev(slab1), dev(W), dev(bias)
    out = dev(torch.zeros(B, Np + 8, 16))  # output rows live in a bigger slab too
    segs = [dict(A=gW.data_ptr(), B=g0.data_ptr() + 4 * 3 * 16, C=out.data_ptr(), M=Np, N=B * 16, K=n0, lda=Nt, ldb=(n0 + 3) * 16, ldc=(Np + 8) * 16),
            dict(A=gW.data_ptr() + 4 * n0, B=g1.data_ptr(), C=out.data_ptr(), M=Np, N=B * 16, K=n1, lda=Nt, ldb=n1 * 16, ldc=(Np + 8) * 16)]
    ws = dev(torch.zeros(splitk * Np * B * 16))
    launch(lib, gemm_desc(L.AM_KC, L.AM_TOKR, L.CM_TOKJ, segs, 0, act=L.ACT_RELU, dims=dims, bias=gb.data_ptr(), bias_on_rows=1, mask_on_rows=1,
                          splitk=splitk, workspace=ws.data_ptr()))
    x = torch.cat([x0, x1], 1).double()
    z = torch.einsum("pn,bne->bpe", W.double(), x) + bias.double()[None, :, None]
    y = z.clamp_min(0)
    y[:, dims:, :] = 0
    close(out[:, :Np, :], y)
    assert float(out[:, Np:, :].abs().max()) == 0.0
    # backward: dz = dy ⊙ [y>0], rows >= dims are zero
    dy = torch.randn(B, Np, 16)
    gdy, gy = dev(dy), out[:, :Np, :].contiguous()
    dz = (dy.double() * (y > 0))
    dx0, dx1 = dev(torch.zeros(B, n0, 16)), dev(torch.full((B, n1, 16), 1.0))
    zs = [dict(A=gW.data_ptr(), B=gdy.data_ptr(), Baux=gy.data_ptr(), C=dx0.data_ptr(), M=n0, N=B * 16, K=dims, lda=Nt, ldb=Np * 16, ldc=n0 * 16, accumulate=0),
          dict(A=gW.data_ptr() + 4 * n0, B=gdy.data_ptr(), Baux=gy.data_ptr(), C=dx1.data_ptr(), M=n1, N=B * 16, K=dims, lda=Nt, ldb=Np * 16, ldc=n1 * 16, accumulate=1)]
    launch(lib, gemm_desc(L.AM_RC, L.AM_TOKR, L.CM_TOKJ, zs, 1))
    dx = torch.einsum("pn,bpe->bne", W.double(), dz)
    close(dx0, dx[:, :n0])
    close(dx1, dx[:, n0:] + 1.0)
    dW = dev(torch.zeros(Np, Nt))
    ws2 = dev(torch.zeros(2 * splitk * Np * max(n0, n1)))
    wsd = [dict(A=gdy.data_ptr(), Aaux=gy.data_ptr(), B=g0.data_ptr() + 4 * 3 * 16, C=dW.data_ptr(), M=Np, N=n0, K=B * 16, lda=Np * 16, ldb=(n0 + 3) * 16, ldc=Nt, Mvalid=dims),
           dict(A=gdy.data_ptr(), Aaux=gy.data_ptr(), B=g1.data_ptr(), C=dW.data_ptr() + 4 * n0, M=Np, N=n1, K=B * 16, lda=Np * 16, ldb=n1 * 16, ldc=Nt, Mvalid=dims)]
    launch(lib, gemm_desc(L.AM_TOKK, L.AM_TOKK, L.CM_PLAIN, wsd, 1, splitk=splitk, workspace=ws2.data_ptr()))
    close(dW, torch.einsum("bpe,bne->pn", dz, x))
    db = dev(torch.zeros(Np))
    r = L.RowsumDesc()
    r.kind, r.mode, r.R, r.K, r.ld, r.rvalid = L.OP_ROWSUM, L.AM_TOKK, Np, B * 16, Np * 16, dims
    r.p, r.aux, r.out = gdy.data_ptr(), gy.data_ptr(), db.data_ptr()
    launch(lib, r)
    close(db, dz.sum((0, 2)))


def test_embedding_gather_bit_exact_and_oob(lib):
    torch.manual_seed(3)
    rows = [5, 1000, 1, 77777]
    Fs, B = len(rows), 300
    tables = [torch.randn(n, 16) for n in rows]
    idx = torch.stack([torch.randint(0, n, (B,)) for n in rows], 1)
    idx[0] = 0
    idx[1] = torch.tensor(rows) - 1
    idx[2] = idx[3]
    gt, gi = [dev(t) for t in tables], dev(idx)
    out, oob = dev(torch.zeros(B, Fs, 16)), dev(torch.zeros(1, dtype=torch.int32))
    d = L.EmbedDesc()
    d.kind, d.B, d.Fs = L.OP_EMBED_GATHER, B, Fs
    d.idx, d.out, d.oob = gi.data_ptr(), out.data_ptr(), oob.data_ptr()
    for f in range(Fs):
        d.table[f], d.rows[f] = gt[f].data_ptr(), rows[f]
    launch(lib, d)
    ref = torch.stack([tables[f][idx[:, f]] for f in range(Fs)], 1)
    assert torch.equal(out.cpu(), ref)  # bit-exact
    assert int(oob.item()) == 0
    # the same gather riding on the per-step staging launch (NASREC_OP_STAGE_INPUTS with desc.gather)
    Fd = 13
    int_src, y_src = dev(torch.randn(B, Fd)), dev(torch.rand(B))
    int_dst, cat_dst, y_dst, lr_dst = dev(torch.zeros(B, Fd)), dev(torch.zeros(B, Fs, dtype=torch.int64)), dev(torch.zeros(B)), dev(torch.zeros(1))
    out2 = dev(torch.zeros(B, Fs, 16))
    st = L.StageDesc()
    st.kind, st.B, st.Fd, st.Fs, st.lr = L.OP_STAGE_INPUTS, B, Fd, Fs, 0.125
    st.int_src, st.int_dst, st.cat_src, st.cat_dst = int_src.data_ptr(), int_dst.data_ptr(), gi.data_ptr(), cat_dst.data_ptr()
    st.y_src, st.y_dst, st.lr_dst = y_src.data_ptr(), y_dst.data_ptr(), lr_dst.data_ptr()
    st.gather = d
    st.gather.out = out2.data_ptr()
    launch(lib, st)
    assert torch.equal(out2.cpu(), ref) and torch.equal(int_dst, int_src) and torch.equal(cat_dst, gi) and torch.equal(y_dst, y_src)
    assert float(lr_dst.item()) == 0.125 and int(oob.item()) == 0
    gi[5, 1] = 1000  # one past the end
    launch(lib, d)
    assert int(oob.item()) == 1


@pytest.mark.parametrize("B", [2048, 600, 256, 77])
def test_rowsparse_adagrad_equals_dense_reference(lib, B):
    """dedup + row-sparse clip·Adagrad == embedding_dense_backward + clip_grad_norm_ + torch.optim.Adagrad (dense).
    B > 256 runs the scan/ballot kernel over several row-blocks, B <= 256 the single-workgroup LDS rank-round kernel."""
    torch.manual_seed(4)
    rows = [4, 50, 3000]
    Fs = 3  # field 0 has ~B/4 duplicates per row
    tables = [torch.randn(n, 16) for n in rows]
    idx = torch.stack([torch.randint(0, n, (B,)) for n in rows], 1)
    dout = torch.randn(B, Fs, 16) * 0.1
    lr, eps, clip = 0.16, 1e-2, 0.05
    # dense reference in fp64, two steps (state carries over)
    ref_t = [torch.nn.Parameter(t.double().clone()) for t in tables]
    opt = torch.optim.Adagrad(ref_t, lr=lr, eps=eps)
    gt, gs = [dev(t.clone()) for t in tables], [dev(torch.zeros_like(t)) for t in tables]
    gi, gd = dev(idx), dev(dout)
    leader, gsum = dev(torch.zeros(B * Fs, dtype=torch.int32)), dev(torch.zeros(B * Fs * 16))
    nb = (B + 255) // 256
    part = dev(torch.zeros(Fs * nb))
    lr_d, coef = dev(torch.tensor([lr])), dev(torch.zeros(2))
    for step in range(2):
        for f in range(Fs):
            ref_t[f].grad = torch.zeros_like(ref_t[f]).index_add_(0, idx[:, f], dout[:, f].double())
        total = torch.nn.utils.clip_grad_norm_(ref_t, clip)
        opt.step()
        dd = L.EmbDedupDesc()
        dd.kind, dd.B, dd.Fs = L.OP_EMB_DEDUP, B, Fs
        dd.idx, dd.dout, dd.leader, dd.gsum, dd.sumsq_partial = gi.data_ptr(), gd.data_ptr(), leader.data_ptr(), gsum.data_ptr(), part.data_ptr()
        launch(lib, dd)
        cc = L.ClipCoefDesc()
        cc.kind, cc.n_a, cc.n_b, cc.max_norm = L.OP_CLIP_COEF, Fs * nb, 0, clip
        cc.partial_a, cc.partial_b, cc.out = part.data_ptr(), part.data_ptr(), coef.data_ptr()
        launch(lib, cc)
        assert abs(float(coef[1]) - float(total)) <= 1e-5 * float(total)
        ar = L.AdagradRowsDesc()
        ar.kind, ar.B, ar.Fs, ar.eps = L.OP_ADAGRAD_ROWS, B, Fs, eps
        ar.idx, ar.leader, ar.gsum, ar.lr, ar.coef = gi.data_ptr(), leader.data_ptr(), gsum.data_ptr(), lr_d.data_ptr(), coef.data_ptr()
        for f in range(Fs):
            ar.table[f], ar.state[f], ar.rows[f] = gt[f].data_ptr(), gs[f].data_ptr(), rows[f]
        launch(lib, ar)
        for f in range(Fs):
            close(gt[f], ref_t[f].data, 1e-5)
    # leaders are exactly the first occurrences
    lead = leader.cpu().view(B, Fs)
    for f in range(Fs):
        first = {}
        for b in range(B):
            first.setdefault(int(idx[b, f]), b)
        want = torch.zeros(B, dtype=torch.int32)
        want[list(first.values())] = 1
        assert torch.equal(lead[:, f], want)


@pytest.mark.parametrize("B", [256, 255, 37, 1])
def test_rowsparse_dedup_sums_duplicates_in_ascending_sample_order(lib, B):
    """one-workgroup dedup (B <= 256) on heavy duplication — every id equal, a 1-row, a 4-row and a 7-row table, one field without
    duplicates: the leader's row is the fp32 sum of its duplicates' rows in ascending sample order, BIT for bit (sixteen lanes per
    leader add one float each; the sequential fp32 loop below is the same chain), leaders = first occurrences, the sum-of-squares
    partial matches, and a second launch gives the same bytes"""
    torch.manual_seed(11)
    Fs = 5
    idx = torch.stack([torch.full((B,), 3), torch.zeros(B, dtype=torch.int64), torch.randint(0, 4, (B,)), torch.randint(0, 7, (B,)),
                       torch.randperm(1000)[:B]], 1)
    dout = torch.randn(B, Fs, 16)
    gi, gd = dev(idx), dev(dout)
    leader, gsum, part = dev(torch.zeros(B * Fs, dtype=torch.int32)), dev(torch.zeros(B * Fs * 16)), dev(torch.full((Fs,), 7.0))
    dd = L.EmbDedupDesc()
    dd.kind, dd.B, dd.Fs = L.OP_EMB_DEDUP, B, Fs
    dd.idx, dd.dout, dd.leader, dd.gsum, dd.sumsq_partial = gi.data_ptr(), gd.data_ptr(), leader.data_ptr(), gsum.data_ptr(), part.data_ptr()
    launch(lib, dd)
    torch.cuda.synchronize()
    lead = leader.cpu().view(B, Fs)
    gs = gsum.cpu().view(B, Fs, 16)
    total = 0.0
    for f in range(Fs):
        first, acc = {}, {}
        for b in range(B):
            i = int(idx[b, f])
            if i in first:
                acc[i] = acc[i] + dout[b, f]  # fp32, ascending b
            else:
                first[i], acc[i] = b, dout[b, f].clone()
        want = torch.zeros(B, dtype=torch.int32)
        want[list(first.values())] = 1
        assert torch.equal(lead[:, f], want), "field %d: leaders are not the first occurrences" % f
        for i, b in first.items():
            assert torch.equal(gs[b, f], acc[i]), "field %d row %d: not the ascending-order fp32 sum" % (f, i)
            total += float(acc[i].double().pow(2).sum())
    assert abs(float(part.double().sum()) - total) <= 1e-5 * total
    g1, p1 = gsum.clone(), part.clone()
    launch(lib, dd)
    torch.cuda.synchronize()
    sel = lead.bool().view(-1)
    assert torch.equal(part, p1) and torch.equal(gsum.view(-1, 16)[sel], g1.view(-1, 16)[sel])


@pytest.mark.parametrize("B", [8192, 20000, 65536])
def test_rowsparse_dedup_at_global_batches_of_a_data_parallel_step(lib, B):
    """the partitioned merge (grid Fs x ceil(B/4096)) at the global batches of BASELINE configs 4 / 5 on 8 GPUs (32 768 and
    65 536 samples): leaders = first occurrences, summed rows == index_add_ in fp64, sum of squares == the dense gradient norm,
    on a 4-row table (every chunk leads every row), mid-size tables and a table larger than the batch (few duplicates)"""
    torch.manual_seed(5)
    rows = [4, 50, 3000, 300000]
    Fs = len(rows)
    idx = torch.stack([torch.randint(0, n, (B,)) for n in rows], 1)
    dout = torch.randn(B, Fs, 16) * 0.1
    gi, gd = dev(idx), dev(dout)
    leader, gsum = dev(torch.zeros(B * Fs, dtype=torch.int32)), dev(torch.zeros(B * Fs * 16))
    nb = (B + 255) // 256
    part = dev(torch.full((Fs * nb,), 7.0))
    flag = dev(torch.zeros(1, dtype=torch.int32))
    dd = L.EmbDedupDesc()
    dd.kind, dd.B, dd.Fs = L.OP_EMB_DEDUP, B, Fs
    dd.idx, dd.dout, dd.leader, dd.gsum, dd.sumsq_partial = gi.data_ptr(), gd.data_ptr(), leader.data_ptr(), gsum.data_ptr(), part.data_ptr()
    dd.overflow = flag.data_ptr()
    launch(lib, dd)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    lead = leader.cpu().view(B, Fs)
    gs = gsum.cpu().view(B, Fs, 16).double()
    total = 0.0
    for f in range(Fs):
        ids, inv = torch.unique(idx[:, f], return_inverse=True)
        dense = torch.zeros(len(ids), 16, dtype=torch.float64).index_add_(0, inv, dout[:, f].double())
        first = torch.full((len(ids),), B, dtype=torch.int64).scatter_reduce_(0, inv, torch.arange(B), reduce="amin")
        want = torch.zeros(B, dtype=torch.int32)
        want[first] = 1
        assert torch.equal(lead[:, f], want), "field %d: leaders are not the first occurrences" % f
        got = gs[first, f]
        assert float((got - dense).abs().max()) <= 1e-5 * max(1.0, float(dense.abs().max())), f
        total += float(dense.pow(2).sum())
    assert abs(float(part.double().sum()) - total) <= 1e-4 * total
    # bit-reproducible: a second launch on the same inputs gives the same bytes
    g1, p1 = gsum.clone(), part.clone()
    launch(lib, dd)
    torch.cuda.synchronize()
    assert torch.equal(leader.cpu().view(B, Fs), lead) and torch.equal(part, p1)
    sel = lead.bool().view(-1)
    assert torch.equal(gsum.view(-1, 16)[sel], g1.view(-1, 16)[sel])


def test_rowsparse_adagrad_skips_out_of_range_ids(lib):
    """an id outside [0, rows) must not be written anywhere (torch raises in the forward pass; the engine flags it in the
    gather and the optimizer leaves memory alone)"""
    B, Fs, rows = 4, 1, 5
    back = dev(torch.full((7, 16), 3.0))  # rows 5, 6 are guard rows behind the 5-row table
    stat = dev(torch.zeros(7, 16))
    idx = dev(torch.tensor([[1], [5], [-1], [1]], dtype=torch.int64))
    dout = dev(torch.ones(B, Fs, 16))
    leader, gsum, part = dev(torch.zeros(B * Fs, dtype=torch.int32)), dev(torch.zeros(B * Fs * 16)), dev(torch.zeros(Fs))
    lr_d, coef = dev(torch.tensor([0.5])), dev(torch.tensor([1.0, 0.0]))
    dd = L.EmbDedupDesc()
    dd.kind, dd.B, dd.Fs = L.OP_EMB_DEDUP, B, Fs
    dd.idx, dd.dout, dd.leader, dd.gsum, dd.sumsq_partial = idx.data_ptr(), dout.data_ptr(), leader.data_ptr(), gsum.data_ptr(), part.data_ptr()
    launch(lib, dd)
    ar = L.AdagradRowsDesc()
    ar.kind, ar.B, ar.Fs, ar.eps = L.OP_ADAGRAD_ROWS, B, Fs, 1e-2
    ar.idx, ar.leader, ar.gsum, ar.lr, ar.coef = idx.data_ptr(), leader.data_ptr(), gsum.data_ptr(), lr_d.data_ptr(), coef.data_ptr()
    ar.table[0], ar.state[0], ar.rows[0] = back.data_ptr(), stat.data_ptr(), rows
    launch(lib, ar)
    torch.cuda.synchronize()
    want = torch.full((7, 16), 3.0)
    want[1] = 3.0 - 0.5 * 2.0 / (2.0 + 1e-2)  # two samples hit row 1: g = 2, state = 4
    close(back, want, 1e-6)
    assert torch.equal(stat[[0, 2, 3, 4, 5, 6]].cpu(), torch.zeros(6, 16))


@pytest.mark.parametrize("B", [256, 77])
def test_fused_optimizer_tail_is_bit_identical_to_the_stand_alone_ops(lib, B):
    """OPT_REDUCE + OPT_APPLY (two launches) == EMB_DEDUP, SUMSQ, CLIP_COEF, ADAGRAD_DENSE, ADAGRAD_ROWS (five launches)"""
    torch.manual_seed(11)
    rows, Fs, n = [4, 50, 3000], 3, 70001
    idx = dev(torch.stack([torch.randint(0, r, (B,)) for r in rows], 1))
    dout = dev(torch.randn(B, Fs, 16) * 0.1)
    g = dev(torch.randn(n) * 0.05)
    lr_d = dev(torch.tensor([0.12]))
    nblk = 37

    def state():
        torch.manual_seed(12)
        return dict(tab=[dev(torch.randn(r, 16)) for r in rows], tst=[dev(torch.rand(r, 16)) for r in rows], p=dev(torch.randn(n)),
                    st=dev(torch.rand(n)), leader=dev(torch.zeros(B * Fs, dtype=torch.int32)), gsum=dev(torch.zeros(B * Fs * 16)),
                    pe=dev(torch.zeros(Fs)), pd=dev(torch.zeros(nblk)), coef=dev(torch.zeros(2)))

    def descs(s):
        dd = L.EmbDedupDesc()
        dd.kind, dd.B, dd.Fs = L.OP_EMB_DEDUP, B, Fs
        dd.idx, dd.dout, dd.leader, dd.gsum, dd.sumsq_partial = idx.data_ptr(), dout.data_ptr(), s["leader"].data_ptr(), s["gsum"].data_ptr(), s["pe"].data_ptr()
        sq = L.SumsqDesc()
        sq.kind, sq.nblocks, sq.n, sq.x, sq.partial = L.OP_SUMSQ, nblk, n, g.data_ptr(), s["pd"].data_ptr()
        cc = L.ClipCoefDesc()
        cc.kind, cc.n_a, cc.n_b, cc.max_norm = L.OP_CLIP_COEF, nblk, Fs, 0.7
        cc.partial_a, cc.partial_b, cc.out = s["pd"].data_ptr(), s["pe"].data_ptr(), s["coef"].data_ptr()
        ad = L.AdagradDenseDesc()
        ad.kind, ad.eps, ad.n = L.OP_ADAGRAD_DENSE, 1e-2, n
        ad.p, ad.g, ad.state, ad.lr, ad.coef = s["p"].data_ptr(), g.data_ptr(), s["st"].data_ptr(), lr_d.data_ptr(), s["coef"].data_ptr()
        ar = L.AdagradRowsDesc()
        ar.kind, ar.B, ar.Fs, ar.eps = L.OP_ADAGRAD_ROWS, B, Fs, 1e-2
        ar.idx, ar.leader, ar.gsum, ar.lr, ar.coef = idx.data_ptr(), s["leader"].data_ptr(), s["gsum"].data_ptr(), lr_d.data_ptr(), s["coef"].data_ptr()
        for f in range(Fs):
            ar.table[f], ar.state[f], ar.rows[f] = s["tab"][f].data_ptr(), s["tst"][f].data_ptr(), rows[f]
        return dd, sq, cc, ad, ar

    a, b = state(), state()
    for d in descs(a):
        launch(lib, d)
    dd, sq, cc, ad, ar = descs(b)
    red = L.OptReduceDesc()
    red.kind, red.dedup, red.sumsq = L.OP_OPT_REDUCE, dd, sq
    app = L.OptApplyDesc()
    app.kind, app.dense_blocks, app.clip, app.dense, app.rows = L.OP_OPT_APPLY, min(2048, (n + 255) // 256), cc, ad, ar
    launch(lib, red)
    launch(lib, app)
    torch.cuda.synchronize()
    assert float(a["coef"][0]) < 1.0  # the clip is active
    for k in ("p", "st", "leader", "pe", "pd", "coef"):
        assert torch.equal(a[k], b[k]), k
    for f in range(Fs):
        assert torch.equal(a["tab"][f], b["tab"][f]) and torch.equal(a["tst"][f], b["tst"][f])
    lead = a["leader"].bool()
    assert torch.equal(a["gsum"].view(-1, 16)[lead], b["gsum"].view(-1, 16)[lead])


@pytest.mark.parametrize("k1", [2, 9, 16, 17, 33, 40, 46, 49, 64])  # (one to four 16-row blocks of the MFMA form, block edges, the maximum)
def test_dot_tri(lib, k1):
    torch.manual_seed(5)
    B = 11
    T = torch.randn(B, k1, 16)
    P = k1 * (k1 - 1) // 2
    gT, out = dev(T), dev(torch.zeros(B, P + 3))
    d = L.DotTriDesc()
    d.kind, d.B, d.k1, d.ld_out = L.OP_DOT_TRI_FWD, B, k1, P + 3
    d.T, d.out = gT.data_ptr(), out.data_ptr()
    launch(lib, d)
    Td = T.double().requires_grad_(True)
    Z = Td @ Td.transpose(1, 2)
    li, lj = torch.tril_indices(k1, k1, offset=-1)
    ref = Z[:, li, lj]
    close(out[:, :P], ref)
    assert bool((out[:, P:] == 0).all()), "columns beyond the triangle were written"
    dout = torch.randn(B, P + 3)
    ref.backward(dout[:, :P].double())
    gdo, dT = dev(dout), dev(torch.zeros(B, k1, 16))
    e = L.DotTriDesc()
    e.kind, e.B, e.k1, e.ld_out = L.OP_DOT_TRI_BWD, B, k1, P + 3
    e.T, e.dout, e.dT = gT.data_ptr(), gdo.data_ptr(), dT.data_ptr()
    launch(lib, e)
    close(dT, Td.grad)


def test_fm(lib):
    torch.manual_seed(6)
    B, N, Ntot = 37, 48, 56
    slab = torch.randn(B, Ntot, 16)
    x = slab[:, :N]
    base = torch.randn(B, 20)
    gx, ix = dev(slab), dev(base.clone())
    d = L.FmDesc()
    d.kind, d.B, d.N, d.ldx, d.ld_ix, d.accumulate = L.OP_FM_FWD, B, N, Ntot * 16, 20, 1
    d.x, d.ix = gx.data_ptr(), ix.data_ptr()
    launch(lib, d)
    xd = x.double().requires_grad_(True)
    ref = xd.sum(1) ** 2 - (xd ** 2).sum(1)
    close(ix[:, :16], base[:, :16].double() + ref)
    dix = torch.randn(B, 20)
    ref.backward(dix[:, :16].double())
    gdix, dx = dev(dix), dev(torch.ones(B, Ntot, 16))
    e = L.FmDesc()
    e.kind, e.B, e.N, e.ldx, e.ld_ix, e.accumulate = L.OP_FM_BWD, B, N, Ntot * 16, 20, 1
    e.x, e.dix, e.dx = gx.data_ptr(), gdix.data_ptr(), dx.data_ptr()
    launch(lib, e)
    close(dx[:, :N], xd.grad + 1.0)
    assert float((dx[:, N:] - 1.0).abs().max()) == 0.0


def _mha_ref(x, p, dims):
    """modules.py:664-686 in fp64"""
    Win, bin_, Wout, bout, l1w, l1b, W1, c1, W2, c2, l2w, l2b = p
    B, N, E = x.shape
    qkv = x @ Win.t() + bin_
    q, k, v = [t.reshape(B, N, 8, 2).permute(0, 2, 1, 3) for t in (qkv[..., :16], qkv[..., 16:32], qkv[..., 32:])]
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(2), -1) @ v
    a = a.permute(0, 2, 1, 3).reshape(B, N, E) @ Wout.t() + bout
    h = torch.nn.functional.layer_norm(a + x, (16,), l1w, l1b, 1e-5)
    f = torch.relu(h @ W1.t() + c1) @ W2.t() + c2
    out = torch.nn.functional.layer_norm(h + f, (16,), l2w, l2b, 1e-5)
    if dims >= 0:
        out = out * (torch.arange(N) < dims).double()[None, :, None]
    return out


@pytest.mark.parametrize("use_saved", [False, True])
@pytest.mark.parametrize("N,dims,B,form", [(16, -1, 13, 0), (64, -1, 13, 0), (48, 32, 13, 0), (7, -1, 13, 0), (64, 40, 1030, 0), (26, -1, 1024, 0),
                                           (64, -1, 13, 4), (48, 32, 9, 4), (7, -1, 5, 4), (17, 16, 6, 4), (1, -1, 3, 4), (33, -1, 258, 4),
                                           # whole blocks of 16 tokens behind the token mask (they act as keys only): round 6
                                           (64, 16, 7, 4), (64, 33, 5, 4), (48, 0, 3, 4), (64, 48, 1027, 0)])
def test_mha_ffn(lib, N, dims, use_saved, B, form):
    """the forward runs with or without saving its per-token state (36 floats per token: attention output, softmax and LayerNorm
    statistics); the backward always consumes a saved state and recomputes the rest (`bwd_form` is ignored since ABI 16)"""
    torch.manual_seed(7 + N)
    shapes = [(48, 16), (48,), (16, 16), (16,), (16,), (16,), (16, 16), (16,), (16, 16), (16,), (16,), (16,)]
    p = [torch.randn(s) * (0.3 if len(s) == 2 else 0.1) for s in shapes]
    p[4] = 0.17 + 0.02 * torch.randn(16)
    p[10] = 0.17 + 0.02 * torch.randn(16)
    x = torch.randn(B, N, 16)
    if dims >= 0:
        x[:, dims:] = 0  # supernet mode zeroes masked tokens before attention (modules.py:653-662)
    gp, gx, out = [dev(t) for t in p], dev(x), dev(torch.zeros(B, N, 16))
    d = L.MhaDesc()
    d.kind, d.B, d.N, d.ldx, d.ldo, d.dims_in_use = L.OP_MHA_FWD, B, N, N * 16, N * 16, dims
    d.x, d.out = gx.data_ptr(), out.data_ptr()
    saved = dev(torch.zeros(B * N * L.MHA_SAVED))
    d.saved = saved.data_ptr() if use_saved else None
    for q in range(12):
        d.params[q] = gp[q].data_ptr()
    launch(lib, d)
    if not use_saved:  # produce the state for the backward with a second, saving launch; outputs must not change
        first = out.clone()
        d.saved = saved.data_ptr()
        launch(lib, d)
        assert torch.equal(first, out)
    pd = [t.double().requires_grad_(True) for t in p]
    xd = x.double().requires_grad_(True)
    ref = _mha_ref(xd, pd, dims)
    close(out, ref, 1e-5)
    dout = torch.randn(B, N, 16)
    ref.backward(dout.double())
    gdo, dx, part = dev(dout), dev(torch.zeros(B, N, 16)), dev(torch.zeros(B, L.MHA_PARAMS))
    e = L.MhaDesc()
    e.kind, e.B, e.N, e.ldx, e.ldo, e.dims_in_use = L.OP_MHA_BWD, B, N, N * 16, N * 16, dims
    e.x, e.dout, e.dx, e.dparams_partial = gx.data_ptr(), gdo.data_ptr(), dx.data_ptr(), part.data_ptr()
    e.saved = saved.data_ptr()
    e.bwd_form = form
    for q in range(12):
        e.params[q] = gp[q].data_ptr()
    launch(lib, e)
    close(dx, xd.grad, 2e-5)
    grads = [dev(torch.zeros(s)) for s in shapes]
    r = L.ReduceRowsDesc()
    r.kind, r.R, r.C, r.ld, r.ndst = L.OP_REDUCE_ROWS, B, L.MHA_PARAMS, L.MHA_PARAMS, 12
    r.in_ = part.data_ptr()
    off = 0
    for q, s in enumerate(shapes):
        n = int(np.prod(s))
        r.dst[q], r.dst_off[q], r.dst_len[q] = grads[q].data_ptr(), off, n
        off += n
    launch(lib, r)
    for q in range(12):
        close(grads[q], pd[q].grad, 2e-5)


@pytest.mark.parametrize("mode", ["dense", "token", "token-b37-d64", "token-b20-d7", "token-b130-d26"])
@pytest.mark.parametrize("act", [L.ACT_NONE, L.ACT_RELU, L.ACT_SILU])
def test_layernorm(lib, mode, act):
    torch.manual_seed(8)
    if mode == "dense":
        R, D, dims = 37, 300, 200
        x = torch.randn(R, D) * 2 + 0.5
    else:
        B, D, dims = {"token": (5, 45, 30), "token-b37-d64": (37, 64, 64), "token-b20-d7": (20, 7, 5), "token-b130-d26": (130, 26, -1)}[mode]
        mode = "token"
        x3 = torch.randn(B, D, 16) * 2 + 0.5  # [B, N', 16]; LN over N' per (b,e)
        R = B * 16
        x = x3.permute(0, 2, 1).reshape(R, D)
    w, b = 1 + 0.1 * torch.randn(D), 0.1 * torch.randn(D)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    u = torch.nn.functional.layer_norm(xd, (D,), wd, bd, 1e-5)
    u = {L.ACT_NONE: u, L.ACT_RELU: u.clamp_min(0), L.ACT_SILU: u * torch.sigmoid(u)}[act]
    ref = u * (torch.arange(D) < (dims if dims >= 0 else D)).double()
    dy = torch.randn(R, D)
    ref.backward(dy.double())

    def to_dev_layout(t):
        if mode == "dense":
            return dev(t)
        return dev(t.reshape(-1, 16, D).permute(0, 2, 1))  # back to [B, N', 16]

    gx, gw, gb, gdy = to_dev_layout(x), dev(w), dev(b), to_dev_layout(dy)
    y, stats, dx = torch.zeros_like(gx), dev(torch.zeros(R * 2)), torch.zeros_like(gx)
    d = L.LayerNormDesc()
    d.kind = L.OP_LAYERNORM_FWD
    d.mode = L.AM_KC if mode == "dense" else L.AM_TOKR
    d.R, d.D = R, D
    d.ldx = d.ldy = D if mode == "dense" else D * 16
    d.act, d.dims_in_use, d.accumulate, d.eps = act, dims, 0, 1e-5
    d.x, d.w, d.b, d.y, d.stats = gx.data_ptr(), gw.data_ptr(), gb.data_ptr(), y.data_ptr(), stats.data_ptr()
    launch(lib, d)

    def from_dev_layout(t):
        if mode == "dense":
            return t
        return t.permute(0, 2, 1).reshape(R, D)

    close(from_dev_layout(y), ref, 1e-5)
    nblk = min((R + 3) // 4, 256) if mode == "dense" else (R + 255) // 256
    part = dev(torch.zeros(nblk * 2 * D))
    d.kind = L.OP_LAYERNORM_BWD
    d.dy, d.dx, d.dwb_partial, d.nblk = gdy.data_ptr(), dx.data_ptr(), part.data_ptr(), nblk
    launch(lib, d)
    close(from_dev_layout(dx), xd.grad, 2e-5)
    pw = part.view(nblk, 2 * D).sum(0).cpu()
    close(pw[:D], wd.grad, 2e-5)
    close(pw[D:], bd.grad, 2e-5)


def test_final_backward_at_large_batch_is_the_elementwise_product_bit_for_bit(lib):
    """B >= 1024: part A runs 256 columns x 16 samples per workgroup (ragged last row block, a column block that ends inside a segment),
    part B eight rows per trip: d seg[b, j] = dlogit[b] * w[off + j] (+ what the buffer held) exactly, dw / dbias against fp64"""
    torch.manual_seed(19)
    B, D, N = 1030, 300, 7
    dl, sl = torch.randn(B, D), torch.randn(B, N, 16)
    K = D + N * 16
    w, bias = torch.randn(1, K) * 0.05, torch.randn(1)
    y = (torch.rand(B) < 0.25).float()
    g = {k: dev(v) for k, v in dict(dl=dl, sl=sl, w=w, bias=bias, y=y).items()}
    logits = dev(torch.zeros(B))
    f = L.FinalDesc()
    f.kind, f.B, f.nseg = L.OP_FINAL_FWD, B, 2
    f.w, f.bias, f.logits = g["w"].data_ptr(), g["bias"].data_ptr(), logits.data_ptr()
    f.seg[0], f.width[0], f.ld[0], f.off[0] = g["dl"].data_ptr(), D, D, 0
    f.seg[1], f.width[1], f.ld[1], f.off[1] = g["sl"].data_ptr(), N * 16, N * 16, D
    launch(lib, f)
    nsplit = 4
    part = dev(torch.full((nsplit * (K + 1),), float("nan")))
    ddl, dsl = dev(torch.zeros(B, D)), dev(torch.ones(B, N, 16))
    loss, dlog = dev(torch.zeros(1)), dev(torch.zeros(B))
    f.kind = L.OP_FINAL_BWD
    f.y, f.loss, f.dlogits_out, f.grad_scale = g["y"].data_ptr(), loss.data_ptr(), dlog.data_ptr(), 1.0 / B
    f.nsplit, f.dw, f.dbias = nsplit, part.data_ptr(), None
    f.dseg[0], f.dseg_accumulate[0] = ddl.data_ptr(), 0
    f.dseg[1], f.dseg_accumulate[1] = dsl.data_ptr(), 1
    launch(lib, f)
    torch.cuda.synchronize()
    assert torch.equal(ddl, dlog[:, None] * g["w"][0, :D][None, :])
    assert torch.equal(dsl.reshape(B, -1), 1.0 + dlog[:, None] * g["w"][0, D:][None, :])
    z = logits.double()
    ref_dlog = (torch.sigmoid(z) - g["y"].double()) / B
    close(dlog, ref_dlog, 1e-6)
    feats = torch.cat([dl, sl.reshape(B, -1)], 1).double()
    sums = part.view(nsplit, K + 1).double().sum(0).cpu()
    close(sums[:K], (dlog.double().cpu()[:, None] * feats).sum(0), 1e-5)
    close(sums[K:], dlog.double().sum().reshape(1), 1e-5)


def test_final_bce_and_dense_optimizer(lib):
    torch.manual_seed(9)
    B, D, N = 300, 128, 48
    dl, sl = torch.randn(B, D), torch.randn(B, N, 16)
    K = D + N * 16
    w, bias = torch.randn(1, K) * 0.05, torch.randn(1)
    y = (torch.rand(B) < 0.25).float()
    g = {k: dev(v) for k, v in dict(dl=dl, sl=sl, w=w, bias=bias, y=y).items()}
    logits, loss, dlog = dev(torch.zeros(B)), dev(torch.zeros(1)), dev(torch.zeros(B))
    f = L.FinalDesc()
    f.kind, f.B, f.nseg = L.OP_FINAL_FWD, B, 2
    f.w, f.bias, f.logits = g["w"].data_ptr(), g["bias"].data_ptr(), logits.data_ptr()
    f.seg[0], f.width[0], f.ld[0], f.off[0] = g["dl"].data_ptr(), D, D, 0
    f.seg[1], f.width[1], f.ld[1], f.off[1] = g["sl"].data_ptr(), N * 16, N * 16, D
    launch(lib, f)
    feats = torch.cat([dl, sl.reshape(B, -1)], 1).double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    z = feats @ wd.t() + bd
    close(logits, z[:, 0])
    bd_ = L.BceDesc()
    bd_.kind, bd_.B, bd_.grad_scale = L.OP_BCE, B, 1.0 / B
    bd_.logits, bd_.y, bd_.loss, bd_.dlogits = logits.data_ptr(), g["y"].data_ptr(), loss.data_ptr(), dlog.data_ptr()
    launch(lib, bd_)
    ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(z[:, 0], y.double())
    close(loss, ref_loss.reshape(1), 1e-6)
    ref_loss.backward()
    ddl, dsl, dw, db = dev(torch.zeros(B, D)), dev(torch.ones(B, N, 16)), dev(torch.zeros(K)), dev(torch.zeros(1))
    f.kind = L.OP_FINAL_BWD
    f.dlogits, f.dw, f.dbias = dlog.data_ptr(), dw.data_ptr(), db.data_ptr()
    f.dseg[0], f.dseg_accumulate[0] = ddl.data_ptr(), 0
    f.dseg[1], f.dseg_accumulate[1] = dsl.data_ptr(), 1
    launch(lib, f)
    close(ddl, feats.grad[:, :D], 1e-5)
    close(dsl.reshape(B, -1), feats.grad[:, D:] + 1.0, 1e-5)
    close(dw, wd.grad[0], 1e-5)
    close(db, bd.grad, 1e-5)
    # BCE fused into the final-logit backward: identical gradients without the separate NASREC_OP_BCE launch
    ddl2, dsl2, dw2, db2 = dev(torch.zeros(B, D)), dev(torch.ones(B, N, 16)), dev(torch.zeros(K)), dev(torch.zeros(1))
    loss2, dlog2 = dev(torch.zeros(1)), dev(torch.zeros(B))
    f2 = L.FinalDesc.from_buffer_copy(f)
    f2.dlogits = None
    f2.y, f2.loss, f2.dlogits_out, f2.grad_scale = g["y"].data_ptr(), loss2.data_ptr(), dlog2.data_ptr(), 1.0 / B
    f2.dw, f2.dbias, f2.dseg[0], f2.dseg[1] = dw2.data_ptr(), db2.data_ptr(), ddl2.data_ptr(), dsl2.data_ptr()
    launch(lib, f2)
    assert torch.equal(dlog2, dlog) and torch.equal(ddl2, ddl) and torch.equal(dsl2, dsl) and torch.equal(dw2, dw) and torch.equal(db2, db)
    close(loss2, ref_loss.reshape(1), 1e-6)
    # NASREC_OP_FINAL_FUSED (forward + the per-sample part of the backward) followed by the rest (FINAL_BWD with dseg_done): the same
    # bits as forward, then backward with the fused loss — logits, feature gradients (overwrite and accumulate), d w, d bias, loss
    logits4, loss4, dlog4 = dev(torch.full((B,), 7.0)), dev(torch.zeros(1)), dev(torch.zeros(B))
    ddl4, dsl4, dw4, db4 = dev(torch.zeros(B, D)), dev(torch.ones(B, N, 16)), dev(torch.zeros(K)), dev(torch.zeros(1))
    f4 = L.FinalDesc.from_buffer_copy(f2)
    f4.kind, f4.logits, f4.loss, f4.dlogits_out = L.OP_FINAL_FUSED, logits4.data_ptr(), loss4.data_ptr(), dlog4.data_ptr()
    f4.dw, f4.dbias, f4.dseg[0], f4.dseg[1] = dw4.data_ptr(), db4.data_ptr(), ddl4.data_ptr(), dsl4.data_ptr()
    launch(lib, f4)
    assert torch.equal(logits4, logits) and torch.equal(ddl4, ddl) and torch.equal(dsl4, dsl)
    assert float(dw4.abs().max()) == 0.0  # (the fused operator leaves the parts that need every sample's logit alone)
    f5 = L.FinalDesc.from_buffer_copy(f4)
    f5.kind, f5.dseg_done = L.OP_FINAL_BWD, 1
    launch(lib, f5)
    assert torch.equal(dw4, dw) and torch.equal(db4, db) and torch.equal(dlog4, dlog) and torch.equal(loss4, loss2)
    assert torch.equal(ddl4, ddl) and torch.equal(dsl4, dsl)  # (dseg_done: the second launch does not touch them again)
    # large-batch form: the batch in nsplit slices -> partial [nsplit, K + 1] (column K = bias gradient), summed by REDUCE_ROWS
    for nsplit in (2, 5):
        part = dev(torch.full((nsplit * (K + 1),), float("nan")))
        f3 = L.FinalDesc.from_buffer_copy(f2)
        ddl3, dsl3, dw3, db3 = dev(torch.zeros(B, D)), dev(torch.ones(B, N, 16)), dev(torch.zeros(K)), dev(torch.zeros(1))
        f3.nsplit, f3.dw, f3.dbias, f3.dseg[0], f3.dseg[1] = nsplit, part.data_ptr(), None, ddl3.data_ptr(), dsl3.data_ptr()
        launch(lib, f3)
        r = L.ReduceRowsDesc()
        r.kind, r.R, r.C, r.ld, r.in_, r.ndst = L.OP_REDUCE_ROWS, nsplit, K + 1, K + 1, part.data_ptr(), 2
        r.dst[0], r.dst_off[0], r.dst_len[0] = dw3.data_ptr(), 0, K
        r.dst[1], r.dst_off[1], r.dst_len[1] = db3.data_ptr(), K, 1
        launch(lib, r)
        assert torch.equal(ddl3, ddl) and torch.equal(dsl3, dsl)
        close(dw3, wd.grad[0], 1e-5)
        close(db3, bd.grad, 1e-5)
    # flat clip + Adagrad vs torch
    n = 100003
    p0, gr = torch.randn(n), torch.randn(n) * 0.01
    P = torch.nn.Parameter(p0.double().clone())
    opt = torch.optim.Adagrad([P], lr=0.16, eps=1e-2)
    gp, gg, gs = dev(p0.clone()), dev(gr), dev(torch.zeros(n))
    part, coef, lr = dev(torch.zeros(64)), dev(torch.zeros(2)), dev(torch.tensor([0.16]))
    for step in range(2):
        P.grad = gr.double().clone()
        total = torch.nn.utils.clip_grad_norm_([P], 0.5)
        opt.step()
        sq = L.SumsqDesc()
        sq.kind, sq.nblocks, sq.n, sq.x, sq.partial = L.OP_SUMSQ, 64, n, gg.data_ptr(), part.data_ptr()
        launch(lib, sq)
        cc = L.ClipCoefDesc()
        cc.kind, cc.n_a, cc.n_b, cc.max_norm = L.OP_CLIP_COEF, 64, 0, 0.5
        cc.partial_a, cc.partial_b, cc.out = part.data_ptr(), part.data_ptr(), coef.data_ptr()
        launch(lib, cc)
        assert abs(float(coef[1]) - float(total)) <= 1e-5 * float(total)
        ad = L.AdagradDenseDesc()
        ad.kind, ad.eps, ad.n = L.OP_ADAGRAD_DENSE, 1e-2, n
        ad.p, ad.g, ad.state, ad.lr, ad.coef = gp.data_ptr(), gg.data_ptr(), gs.data_ptr(), lr.data_ptr(), coef.data_ptr()
        launch(lib, ad)
        close(gp, P.data, 1e-5)


def test_chunk_table_restricts_zero_grad_norm_and_adagrad_to_a_path(lib):
    """A sampled path's arena ranges as a chunk table (NASREC_OP_CONST_I64 writes it from the kernel arguments): memset, sum of
    squares and Adagrad touch exactly those ranges — everything else (stale gradients of other paths included) is bit-untouched,
    inside the ranges the update is the whole-buffer kernel's, bit for bit."""
    from nasrec_amd import plan as P
    torch.manual_seed(12)
    n = 1_000_000
    ranges = [(0, 128), (128, 4), (4096, 70001), (200000, 65536 * 3 + 8), (999000, 999)]
    flat = P.path_chunks(ranges)
    assert flat[:2] == [0, 132] and sum(flat[1::2]) == sum(r[1] for r in ranges) and max(flat[1::2]) <= L.CHUNK_ELEMS
    nch = len(flat) // 2
    tab = dev(torch.zeros(len(flat), dtype=torch.int64))
    for d in P.const_i64_descs(tab.data_ptr(), flat):
        launch(lib, d)
    assert tab.cpu().tolist() == flat
    big = list(range(1000))
    tab2 = dev(torch.zeros(1000, dtype=torch.int64))
    descs = P.const_i64_descs(tab2.data_ptr(), big)
    assert len(descs) == 3
    for d in descs:
        launch(lib, d)
    assert tab2.cpu().tolist() == big
    mask = torch.zeros(n, dtype=torch.bool)
    for off, m in ranges:
        mask[off:off + m] = True
    mask = mask.cuda()
    g0 = dev(torch.randn(n))
    # memset
    g = g0.clone()
    ms = P.memset_desc(g)
    ms.chunks, ms.nchunks = tab.data_ptr(), nch
    launch(lib, ms)
    assert torch.equal(g[~mask], g0[~mask]) and float(g[mask].abs().max()) == 0.0
    # sum of squares over the ranges only
    part = dev(torch.zeros(256))
    sq = L.SumsqDesc()
    sq.kind, sq.nblocks, sq.n, sq.x, sq.partial = L.OP_SUMSQ, min(256, nch), n, g0.data_ptr(), part.data_ptr()
    sq.chunks, sq.nchunks = tab.data_ptr(), nch
    launch(lib, sq)
    want = float(g0[mask].double().pow(2).sum())
    assert abs(float(part.double().sum()) - want) <= 1e-6 * want
    # Adagrad: ranges == the whole-buffer kernel, outside untouched
    coef, lr = dev(torch.tensor([0.37, 0.0])), dev(torch.tensor([0.16]))
    p0, s0 = dev(torch.randn(n)), dev(torch.rand(n))
    outs = []
    for chunked in (False, True):
        p, st = p0.clone(), s0.clone()
        ad = L.AdagradDenseDesc()
        ad.kind, ad.eps, ad.n = L.OP_ADAGRAD_DENSE, 1e-2, n
        ad.p, ad.g, ad.state, ad.lr, ad.coef = p.data_ptr(), g0.data_ptr(), st.data_ptr(), lr.data_ptr(), coef.data_ptr()
        if chunked:
            ad.chunks, ad.nchunks = tab.data_ptr(), nch
        launch(lib, ad)
        outs.append((p, st))
    torch.cuda.synchronize()
    (pw, sw), (pc, sc) = outs
    assert torch.equal(pc[mask], pw[mask]) and torch.equal(sc[mask], sw[mask])
    assert torch.equal(pc[~mask], p0[~mask]) and torch.equal(sc[~mask], s0[~mask])
    assert not torch.equal(pw[~mask], p0[~mask])


def test_gate_backward_and_copy_segments(lib):
    torch.manual_seed(10)
    B, D, wr = 21, 32, 13
    dout, gate, R = torch.randn(B, D), torch.rand(B, D), torch.randn(B, wr)
    g = {k: dev(v) for k, v in dict(dout=dout, g=gate, R=R).items()}
    dz, dR = dev(torch.zeros(B, D)), dev(torch.ones(B, wr))
    e = L.GateBwdDesc()
    e.kind, e.B, e.D, e.ld_dout, e.ld_g, e.ld_dz = L.OP_GATE_BWD, B, D, D, D, D
    e.dout, e.g, e.dz, e.nseg = g["dout"].data_ptr(), g["g"].data_ptr(), dz.data_ptr(), 1
    e.r_ptr[0], e.dr_ptr[0], e.r_off[0], e.r_width[0], e.r_ld[0], e.dr_accumulate[0] = g["R"].data_ptr(), dR.data_ptr(), 0, wr, wr, 1
    launch(lib, e)
    Rp = torch.cat([R, torch.zeros(B, D - wr)], 1).double()
    close(dz, dout.double() * Rp * gate.double() * (1 - gate.double()))
    close(dR, 1.0 + (dout.double() * gate.double())[:, :wr])
    # copy / reverse-copy
    a, b2 = torch.randn(B, 5), torch.randn(B, 9, 16)
    ga, gb = dev(a), dev(b2)
    dst = dev(torch.zeros(B, 5 + 7 + 9 * 16))
    c = L.CopySegsDesc()
    c.kind, c.B, c.nseg, c.ld_dst, c.accumulate, c.reverse = L.OP_COPY_SEGS, B, 2, dst.shape[1], 0, 0
    c.dst = dst.data_ptr()
    c.seg[0], c.width[0], c.ld[0], c.off[0] = ga.data_ptr(), 5, 5, 0
    c.seg[1], c.width[1], c.ld[1], c.off[1] = gb.data_ptr(), 9 * 16, 9 * 16, 12
    launch(lib, c)
    ref = torch.cat([a, torch.zeros(B, 7), b2.reshape(B, -1)], 1)
    assert torch.equal(dst.cpu(), ref)
    c.reverse = 1
    c.seg_accumulate[0], c.seg_accumulate[1] = 1, 0
    launch(lib, c)
    assert torch.equal(ga.cpu(), a + a) and torch.equal(gb.cpu(), b2)


def test_gemm_splitk_stress(lib):
    """split-K partial slabs + fixed-order second pass: replay the SAME descriptor many times with changing operands and
    require bit-identical results for identical inputs (fixed summation order)."""
    torch.manual_seed(11)
    B, N, K, S = 256, 200, 1565, 7
    W, bias = dev(torch.randn(N, K) * 0.05), dev(torch.randn(N))
    x, y = dev(torch.zeros(B, K)), dev(torch.zeros(B, N))
    ws = dev(torch.zeros(S * B * N))
    d = gemm_desc(L.AM_KC, L.AM_KC, L.CM_PLAIN, [dict(A=x.data_ptr(), B=W.data_ptr(), C=y.data_ptr(), M=B, N=N, K=K, lda=K, ldb=K, ldc=N)],
                  0, act=L.ACT_RELU, bias=bias.data_ptr(), splitk=S, workspace=ws.data_ptr())
    first = None
    for it in range(25):
        xin = torch.randn(B, K) if it % 5 else torch.ones(B, K) * 0.01
        x.copy_(xin)
        torch.cuda.synchronize()
        for _ in range(3):  # back-to-back launches without host sync in between
            L.check(lib.nasrec_launch(None, C.addressof(d)))
        torch.cuda.synchronize()
        ref = (xin.double() @ W.double().cpu().t() + bias.double().cpu()).clamp_min(0)
        close(y, ref)
        if it % 5 == 0:
            if first is None:
                first = y.clone()
            else:
                assert torch.equal(first, y)  # same inputs -> same bits


def test_splitk_second_passes_of_several_launches_as_one(lib):
    """NASREC_OP_SPLITK_EPILOGUES: three split-K weight-gradient launches (ReLU-mask operand, bias column, accumulation, different
    split factors and problem counts) write only their slabs (defer_second_pass); one launch then runs all three second passes —
    bit-identical to every launch running its own"""
    torch.manual_seed(31)
    B = 256
    specs = [([(96, 130, 1), (40, 37, 0)], 4), ([(64, 64, 0)], 7), ([(13, 14, 1), (200, 129, 0), (32, 16, 0)], 3)]  # [(nout, nin, ones)], S
    keep = []

    def build(defer):
        descs, outs = [], []
        g = torch.Generator().manual_seed(5)
        for probs, S in specs:
            segs = []
            for q, (nout, nin, ones) in enumerate(probs):
                dy, y, x = torch.randn(B, nout, generator=g) * 0.3, torch.randn(B, nout, generator=g), torch.randn(B, nin, generator=g)
                dw, db = dev(torch.randn(nout, nin, generator=g)), dev(torch.zeros(nout))
                t = [dev(dy), dev(y), dev(x)]
                keep.extend(t + [dw, db])
                segs.append(dict(A=t[0].data_ptr(), Aaux=t[1].data_ptr(), B=t[2].data_ptr(), C=dw.data_ptr(), M=nout, N=nin + ones, K=B, lda=nout, ldb=nin,
                                 ldc=nin, Mvalid=nout - 3, accumulate=q % 2, ones_col=ones, rowsum=db.data_ptr() if ones else None))
                outs += [dw, db]
            Mm, Nm = max(sd["M"] for sd in segs), max(sd["N"] for sd in segs)
            ws = dev(torch.full((S * Mm * Nm * len(segs),), float("nan")))
            keep.append(ws)
            d = gemm_desc(L.AM_RC, L.AM_RC, L.CM_PLAIN, segs, 1, splitk=S, workspace=ws.data_ptr())
            d.defer_second_pass = int(defer)
            descs.append(d)
        return descs, outs

    plain, want = build(False)
    for d in plain:
        launch(lib, d)
    deferred, got = build(True)
    for d in deferred:
        launch(lib, d)
    e = L.SplitkEpiloguesDesc()
    e.kind, e.n = L.OP_SPLITK_EPILOGUES, len(deferred)
    for q, d in enumerate(deferred):
        C.memmove(C.addressof(e.g[q]), C.addressof(d), C.sizeof(L.GemmDesc))
    launch(lib, e)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert float(want[0].abs().max()) > 0
    bad = L.SplitkEpiloguesDesc()
    bad.kind, bad.n = L.OP_SPLITK_EPILOGUES, 4
    with pytest.raises(L.EngineError):
        launch(lib, bad)


def test_gemm_bias_gradient_as_ones_column(lib):
    """db = column sums of dz come out of the dW product as a virtual ones-column (dense and token bindings)"""
    torch.manual_seed(12)
    B, N, K, dims = 130, 64, 96, 50   # N a multiple of the tile width: the virtual column opens a new tile
    dy, y, x = torch.randn(B, N), torch.randn(B, N), torch.randn(B, K)
    g = {k: dev(v) for k, v in dict(dy=dy, y=y, x=x).items()}
    dW, db = dev(torch.zeros(N, K)), dev(torch.zeros(N))
    ws = dev(torch.zeros(4 * N * (K + 1)))
    seg = dict(A=g["dy"].data_ptr(), Aaux=g["y"].data_ptr(), B=g["x"].data_ptr(), C=dW.data_ptr(), M=N, N=K + 1, K=B, lda=N, ldb=K, ldc=K,
               Mvalid=dims, ones_col=1)
    for S in (1, 4):
        dW.zero_(); db.zero_()
        launch(lib, gemm_desc(L.AM_RC, L.AM_RC, L.CM_PLAIN, [seg], 1, splitk=S, workspace=ws.data_ptr(), rowsum_out=db.data_ptr()))
        dz = (dy * (y > 0)).double()
        dz[:, dims:] = 0
        close(dW, dz.t() @ x.double())
        close(db, dz.sum(0))
