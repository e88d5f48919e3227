import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from nasrec_amd import _lib as L
lib = L.load()
torch.manual_seed(0)
def run(B, M, specs, label):
    # specs: list of (K, lda_extra, with_aux)
    N = B * 16
    out = torch.zeros(B, M, 16, device="cuda")
    d = L.GemmDesc(); d.kind = L.OP_GEMM
    d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use, d.splitk = L.AM_RC, L.AM_TOKR, L.CM_TOKJ, len(specs), 0, -1, 1
    want = torch.zeros(B, M, 16, dtype=torch.float64, device="cuda")
    keep = []
    for q, (K, extra, aux) in enumerate(specs):
        lda = M + extra
        W = torch.randn(K, lda, device="cuda")          # A(i,k) = W[k*lda + i]
        dz = torch.randn(B, K + 3, 16, device="cuda")   # B(j,k) = dz[(j>>4)*ldb + k*16 + (j&15)]
        y = torch.randn(B, K + 3, 16, device="cuda")
        keep += [W, dz, y]
        s = d.seg[q]
        s.A, s.B, s.C = W.data_ptr(), dz.data_ptr(), out.data_ptr()
        s.M, s.N, s.K, s.lda, s.ldb, s.ldc, s.Mvalid = M, N, K, lda, dz.stride(0), M * 16, M
        g = dz[:, :K].double()
        if aux:
            s.Baux = y.data_ptr()
            g = g * (y[:, :K] > 0).double()
        want += torch.einsum("ki,bke->bie", W[:, :M].double(), g)
    L.check(lib.nasrec_launch(None, C.addressof(d))); torch.cuda.synchronize()
    err = float((out.double() - want).abs().max())
    print("%-40s err %.3e" % (label, err))
for B in (8, 256):
    run(B, 10, [(48, 0, False)], "B=%d one seg" % B)
    run(B, 10, [(48, 0, False), (45, 0, False)], "B=%d two segs same lda" % B)
    run(B, 10, [(48, 192, False), (45, 128, False)], "B=%d two segs different lda" % B)
    run(B, 10, [(48, 0, True), (45, 0, True)], "B=%d two segs both aux" % B)
    run(B, 10, [(48, 0, True), (45, 0, False)], "B=%d two segs mixed aux" % B)
    run(B, 26, [(32, 5, False), (39, 7, False), (64, 0, False)], "B=%d three segs M=26" % B)
