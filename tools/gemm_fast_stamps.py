#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock stamps of one workgroup of gemm_fast_kernel (library built with -DFT_STAMPS).
Phases per k-tile: [top -> loads issued] [MFMA phase] [park in LDS] [barrier]."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from nasrec_amd import _lib as L
import gemm_fast_bench as G
lib = L.load()
for kind, M, N, K in (("F", 4096, 1024, 1024), ("F", 8192, 1024, 1024), ("DW", 1024, 1024, 4096)):
    d, Cout, ref, ven, keep = G.desc(kind, M, N, K)
    stamps = torch.zeros(512, dtype=torch.int64, device="cuda")
    for _ in range(3):
        L.check(lib.nasrec_launch(G.st, C.addressof(d)))
    d.counters = stamps.data_ptr()
    L.check(lib.nasrec_launch(G.st, C.addressof(d)))
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    n = int((s != 0).sum())
    s = s[:n]
    it = (n - 1) // 4
    a = s[:4 * it].reshape(it, 4)
    issue, mfma, park = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
    barrier = np.append(a[1:, 0], s[4 * it]) - a[:, 3]
    tot = s[4 * it] - s[0]
    print("%s %dx%dx%d: %d k-tiles, %d cycles total, per tile: issue %.0f  mfma %.0f  park %.0f  barrier %.0f  (sum %.0f; pure MFMA would be 4096)" % (
        kind, M, N, K, it, tot, issue.mean(), mfma.mean(), park.mean(), barrier.mean(), tot / it))
    print("   first 6 tiles:", [tuple(int(x) for x in r) for r in np.stack([issue, mfma, park, barrier], 1)[:6]])
