#!/usr/bin/env python3
"""fp32 GEMM of the engine against the vendor library (torch.matmul -> hipBLASLt/rocBLAS, TF32-like modes off) on
supernet-sized products: y[M,N] = x[M,K] . W[N,K]^T, timed with HIP events.  Context for DESIGN.md's MFMA-utilisation numbers."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nasrec_amd import _lib as L

lib = L.load()
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
for (M, N, K) in [(4096, 1024, 1024), (4096, 1024, 4096), (4096, 768, 2048), (8192, 1024, 1024), (256, 768, 1565)]:
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    y = torch.empty(M, N, device=dev)
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use, d.splitk = L.OP_GEMM, L.AM_KC, L.AM_KC, L.CM_PLAIN, 1, 0, -1, 1
    s = d.seg[0]
    s.A, s.B, s.C, s.M, s.N, s.K, s.lda, s.ldb, s.ldc, s.Mvalid = x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, M
    ms = bench.time_desc(lib, L, st, d, iters=30)
    ref = x @ w.t()
    err = float((y - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        torch.matmul(x, w.t(), out=ref)
    e0.record()
    for _ in range(30):
        torch.matmul(x, w.t(), out=ref)
    e1.record()
    torch.cuda.synchronize()
    msv = e0.elapsed_time(e1) / 30
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d  engine %.1f us = %.1f TFLOP/s   vendor %.1f us = %.1f TFLOP/s   rel.err %.1e" % (
        M, N, K, ms * 1e3, fl / ms / 1e9, msv * 1e3, fl / msv / 1e9, err))
