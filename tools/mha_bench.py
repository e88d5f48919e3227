#!/usr/bin/env python3
"""Stand-alone launch times of the fused Transformer kernels (forward with saved state, backward in the 4-wave form the batch-256 step and
the large-batch plans run) at the shapes of the bench configurations.  NASREC_HIP_LIB selects a variant build for A/B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from nasrec_amd import _lib as L

lib = L.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
st = torch.cuda.current_stream().cuda_stream


def launch(d):
    L.check(lib.nasrec_launch(C.c_void_p(st), C.addressof(d)))


shapes = [(48, 16), (48,), (16, 16), (16,), (16,), (16,), (16, 16), (16,), (16, 16), (16,), (16,), (16,)]
for B, N, dims in ((256, 64, -1), (256, 48, -1), (256, 8, -1), (4096, 64, -1), (4096, 64, 48), (4096, 64, 32), (4096, 64, 16), (4096, 26, -1), (8192, 10, -1)):
    torch.manual_seed(0)
    p = [(torch.randn(s) * (0.3 if len(s) == 2 else 0.1)).to(dev) for s in shapes]
    x, out, dout = torch.randn(B, N, 16, device=dev), torch.zeros(B, N, 16, device=dev), torch.randn(B, N, 16, device=dev)
    if dims >= 0:
        x[:, dims:] = 0  # (the supernet zeroes masked tokens in front of the attention: modules.py:653-662)
    dx, part, saved = torch.zeros(B, N, 16, device=dev), torch.zeros(B, L.MHA_PARAMS, device=dev), torch.zeros(B * N * L.MHA_SAVED, device=dev)
    f, b = L.MhaDesc(), L.MhaDesc()
    for d, kind in ((f, L.OP_MHA_FWD), (b, L.OP_MHA_BWD)):
        d.kind, d.B, d.N, d.ldx, d.ldo, d.dims_in_use = kind, B, N, N * 16, N * 16, dims
        d.x, d.out, d.dout, d.dx, d.dparams_partial, d.saved = x.data_ptr(), out.data_ptr(), dout.data_ptr(), dx.data_ptr(), part.data_ptr(), saved.data_ptr()
        d.bwd_form = 4
        for q in range(12):
            d.params[q] = p[q].data_ptr()
    res = []
    for d in (f, b):
        for _ in range(20):
            launch(d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for _ in range(reps):
            launch(d)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / reps)
    print("B=%5d N=%2d dims=%2d: forward %7.2f us   backward %7.2f us   (checksums %.6e %.6e %.6e)" % (B, N, dims, res[0], res[1], float(out.double().sum()), float(dx.double().sum()), float(part.double().sum())))
