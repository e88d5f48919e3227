#!/usr/bin/env python3
"""The single-pass batch-256 GEMM (csrc/gemm_kslice.hip) on aligned and unaligned operand rows: y[256, 768] = x[256, K] W[768, K]^T with
K = 1565 (rows start on 4-byte boundaries only), K = 1568 (16-byte aligned rows) and K = 1565 in a leading dimension of 1568, against
the vendor library.  NASREC_KSLICE_FORM=staged selects the LDS-staged form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nasrec_amd import _lib as L

lib = L.load()
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
M, N = 256, 768
for (K, ld, koff) in [(1565, 1565, 0), (1568, 1568, 0), (1565, 1568, 0), (1564, 1568, 1), (780, 780, 0), (780, 784, 0), (3136, 3136, 0)]:
    xf, wf = torch.randn(M, ld, device=dev), torch.randn(N, ld, device=dev)
    x, w = xf[:, koff:koff + K], wf[:, koff:koff + K]
    y = torch.empty(M, N, device=dev)
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use, d.splitk = L.OP_GEMM, L.AM_KC, L.AM_KC, L.CM_PLAIN, 1, 0, -1, 1
    s = d.seg[0]
    s.A, s.B, s.C, s.M, s.N, s.K, s.lda, s.ldb, s.ldc, s.Mvalid = x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, ld, ld, N, M
    ms = bench.time_desc(lib, L, st, d, iters=100)
    ref = x @ w.t()
    err = float((y - ref).abs().max() / ref.abs().max())
    xc, wc = x.contiguous(), w.contiguous()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        torch.matmul(xc, wc.t(), out=ref)
    e0.record()
    for _ in range(100):
        torch.matmul(xc, wc.t(), out=ref)
    e1.record()
    torch.cuda.synchronize()
    msv = e0.elapsed_time(e1) / 100
    fl = 2.0 * M * N * K
    print("K=%d ld=%d first column %d: engine %.2f us = %.1f TFLOP/s   vendor (contiguous K) %.2f us   rel.err %.1e" % (
        K, ld, koff, ms * 1e3, fl / ms / 1e9, msv * 1e3, err))

# the dominant launch of the Criteo best-1shot step: four K-segments (13 + 768 + 768 + 16), x segments are separate tensors, W one [768, 1565] matrix
for widths in ([13, 768, 768, 16], [768, 13], [16, 16, 1000], [130, 700]):
    K = sum(widths)
    xs = [torch.randn(M, wd, device=dev) for wd in widths]
    w = torch.randn(N, K, device=dev)
    bias = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use, d.splitk = L.OP_GEMM, L.AM_KC, L.AM_KC, L.CM_PLAIN, len(widths), 0, -1, 1
    d.bias, d.act = bias.data_ptr(), L.ACT_RELU
    off = 0
    for q, wd in enumerate(widths):
        s = d.seg[q]
        s.A, s.B, s.C, s.M, s.N, s.K, s.lda, s.ldb, s.ldc, s.Mvalid = xs[q].data_ptr(), w.data_ptr() + 4 * off, y.data_ptr(), M, N, wd, wd, K, N, M
        off += wd
    ms = bench.time_desc(lib, L, st, d, iters=100)
    ref = torch.relu(torch.cat(xs, 1) @ w.t() + bias)
    err = float((y - ref).abs().max() / ref.abs().max())
    print("K=%s + bias + relu: engine %.2f us = %.1f TFLOP/s   rel.err %.1e" % ("+".join(map(str, widths)), ms * 1e3, 2.0 * M * N * K / ms / 1e9, err))
