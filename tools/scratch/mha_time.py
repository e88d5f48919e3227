"""MHA forward / backward at B=4096 and B=256, N=64: with and without the saved state (what bounds the kernels?)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nasrec_amd import _lib as L
lib = L.load()
torch.manual_seed(0)
lens = [768, 48, 256, 16, 16, 16, 256, 16, 256, 16, 16, 16]
gp = [torch.randn(n, device="cuda") * 0.2 for n in lens]
sp = torch.cuda.current_stream().cuda_stream
for B in (4096, 256):
    N = 64
    x, out = torch.randn(B, N, 16, device="cuda"), torch.zeros(B, N, 16, device="cuda")
    saved = torch.zeros(B * N * L.MHA_SAVED, device="cuda")
    d = L.MhaDesc()
    d.kind, d.B, d.N, d.ldx, d.ldo, d.dims_in_use = L.OP_MHA_FWD, B, N, N * 16, N * 16, 48
    d.x, d.out = x.data_ptr(), out.data_ptr()
    for q in range(12):
        d.params[q] = gp[q].data_ptr()
    for sv in (None, saved):
        d.saved = sv.data_ptr() if sv is not None else None
        print("B=%d fwd saved=%s: %.1f us" % (B, sv is not None, bench.time_desc(lib, L, sp, d, iters=30) * 1e3))
    dout, dx, part = torch.randn(B, N, 16, device="cuda"), torch.zeros(B, N, 16, device="cuda"), torch.zeros(B * L.MHA_PARAMS, device="cuda")
    e = L.MhaDesc()
    e.kind, e.B, e.N, e.ldx, e.ldo, e.dims_in_use = L.OP_MHA_BWD, B, N, N * 16, N * 16, 48
    e.x, e.dout, e.dx, e.dparams_partial, e.saved = x.data_ptr(), dout.data_ptr(), dx.data_ptr(), part.data_ptr(), saved.data_ptr()
    for q in range(12):
        e.params[q] = gp[q].data_ptr()
    print("B=%d bwd: %.1f us" % (B, bench.time_desc(lib, L, sp, e, iters=30) * 1e3))
