import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_supernet_fullsize_gpu import _build, _jsonable
from nasrec_amd import _lib as L, plan as P
lib = L.load()
model, c, ds, tables, int_x, cat_x, y = _build("cfg5_kdd_autoctr_b8192", seed=1)
eng = model._engine
for step in range(3):
    ch = _jsonable(model._resolve_choice(None))
cp = eng.compile(ch, c["B"], train=True)
sp = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
nbad = 0
for idx, d0 in enumerate(cp.bwd.descs):
    if not isinstance(d0, L.GemmDesc) or not (d0.zmode and d0.amode == 1 and d0.bmode == 1 and d0.cmode == 0):
        continue
    d = L.GemmDesc.from_buffer_copy(d0)
    keep, want = [], []
    Mx = max(d.seg[q].M for q in range(d.nseg)); Nx = max(d.seg[q].N for q in range(d.nseg))
    S = max(1, d.splitk)
    ws = torch.full((S * Mx * Nx * d.nseg + 16,), float("nan"), device="cuda")
    d.workspace = ws.data_ptr()
    for q in range(d.nseg):
        s = d.seg[q]
        if not s.A:
            continue
        M, N, K = s.M, s.N, s.K
        Nr = N - 1 if s.ones_col else N
        A = torch.randn(K * s.lda + M + 64, device="cuda")
        B = torch.randn(K * s.ldb + Nr + 64, device="cuda")
        Cb = torch.randn(M * s.ldc + N + 64, device="cuda")
        rs = torch.zeros(M + 8, device="cuda")
        C0 = Cb.clone()
        s.A, s.B, s.C = A.data_ptr(), B.data_ptr(), Cb.data_ptr()
        if s.ones_col:
            s.rowsum = rs.data_ptr()
        Am = torch.as_strided(A, (K, M), (s.lda, 1)).double()
        Bm = torch.as_strided(B, (K, Nr), (s.ldb, 1)).double()
        ref = Am.t() @ Bm
        keep.append((A, B, Cb, rs, C0, ref, Am.sum(0), q, (M, N, K, s.lda, s.ldb, s.ldc, s.ones_col, s.accumulate, s.Mvalid)))
    L.check(lib.nasrec_launch(sp, C.addressof(d)))
    torch.cuda.synchronize()
    msgs = []
    for A, B, Cb, rs, C0, ref, rsum, q, shp in keep:
        M, N, K, lda, ldb, ldc, ones, acc, mv = shp
        Nr = N - 1 if ones else N
        got = torch.as_strided(Cb, (M, Nr), (ldc, 1)).double()
        base = torch.as_strided(C0, (M, Nr), (ldc, 1)).double() if acc else 0
        err = float((got - (ref + base)).abs().max())
        e2 = float((rs[:M].double() - rsum).abs().max()) if ones else 0.0
        if err > 2e-3 or e2 > 2e-3:
            msgs.append("q%d %s err %.3g rowsum err %.3g" % (q, shp, err, e2))
    tag = "BAD" if msgs else "ok "
    nbad += bool(msgs)
    print(tag, idx, P.gemm_kernel_name(d), "S=%d" % d.splitk, [(d.seg[q].M, d.seg[q].N, d.seg[q].K) for q in range(d.nseg)], msgs)
print("bad launches:", nbad)
