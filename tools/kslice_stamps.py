#!/usr/bin/env python3
"""In-kernel time stamps of the single-pass batch-256 GEMM (csrc/gemm_kslice.hip built with -DKS_STAMPS: tools/build_variant.sh stamps
-DKS_STAMPS; NASREC_HIP_LIB=nasrec_amd/lib/variants/stamps.so).  Wave 0 of every workgroup writes s_memtime at: 0 entry, 4 k-loop done, 5 slices parked in LDS, 6 epilogue stored; 7 = cycles wave 0 waited at barriers; feeder wave 8 writes its waiting / barrier /
issue cycles into 1..3.  Printed: medians over workgroups of the differences, in us (clock from the span of one launch vs its event time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from nasrec_amd import _lib as L

lib = L.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
M, N = 256, 768
for K in (1565, 780):
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    y = torch.empty(M, N, device=dev)
    ws = torch.zeros(192 * 64, device=dev)
    d = L.GemmDesc()
    d.kind, d.amode, d.bmode, d.cmode, d.nseg, d.zmode, d.dims_in_use, d.splitk = L.OP_GEMM, L.AM_KC, L.AM_KC, L.CM_PLAIN, 1, 0, -1, 1
    d.workspace = ws.data_ptr()
    s = d.seg[0]
    s.A, s.B, s.C, s.M, s.N, s.K, s.lda, s.ldb, s.ldc, s.Mvalid = x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, M
    ms = bench.time_desc(lib, L, st, d, iters=50)
    torch.cuda.synchronize()
    t = ws.view(torch.int32).cpu().numpy().astype(np.int64).reshape(192, 64) & 0xffffffff
    rel = (t[:, 1:7] - t[:, :1]) & 0xffffffff
    med = np.median(t, 0).astype(np.int64)
    w = med[16:64].reshape(16, 3)
    print("K=%d: %.2f us per launch; medians over workgroups: k-loop done %d cycles after entry, slices parked %d, epilogue stored %d" % (
        K, ms * 1e3, np.median(rel[:, 3]), np.median(rel[:, 4]), np.median(rel[:, 5])))
    print("   multiplier waves 0-7 (cycles at barriers, cycles from the first barrier to the end of the loop):", [(int(w[i, 1]), int(w[i, 2])) for i in range(8)])
    print("   feeder waves 8-15 (cycles waiting for own pieces, at barriers, issuing):", [(int(w[i, 0]), int(w[i, 1]), int(w[i, 2])) for i in range(8, 16)])
