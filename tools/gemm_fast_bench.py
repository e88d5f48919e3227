#!/usr/bin/env python3
"""Throughput-regime GEMM (csrc/gemm_fast.hip) on the supernet's fp32 products: correctness against an fp64 product and speed
against the general kernel (NASREC_GEMM_FAST_MIN_TILES forced high via the `general` desc trick: Aaux = ones) and against the
vendor library (torch.matmul fp32).  Bindings: F  y = x W^T (KC/KC), DX dx = dy W (KC/RC), DW dW = dy^T x (RC/RC, split-K).

    python tools/gemm_fast_bench.py [--quick]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nasrec_amd import _lib as L  # noqa: E402
from nasrec_amd import plan as P  # noqa: E402

lib = L.load()
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream


class _Ctx:
    def __init__(self):
        self.keep = []
        self.B = 4096            # (plan.gemm_descs reads the batch regime and the shared split-K workspace from its context)
        self.sk_workspace = None
        self.shape_only = False

    def alloc(self, n):
        t = torch.empty(int(n), dtype=torch.float32, device=dev)
        self.keep.append(t)
        return t


def desc(kind, M, N, K, ldw=None, general=False, aux_keep=[]):
    """kind F: C[M,N] = A[M,K] W[N,K]^T ; DX: C[M,N] = A[M,K] W[K,N] ; DW: C[M,N] = A[K,M]^T X[K,N]"""
    ctx = _Ctx()
    if kind == "F":
        ldw = ldw or K
        A = torch.randn(M, K, device=dev)
        Wfull = torch.randn(N, ldw, device=dev)
        sd = dict(A=A.data_ptr(), B=Wfull.data_ptr() + 4 * (ldw - K), C=None, M=M, N=N, K=K, lda=K, ldb=ldw, ldc=N)
        ref = lambda: A.double() @ Wfull[:, ldw - K:].double().t()
        ven = lambda out: torch.matmul(A, Wfull[:, ldw - K:].t(), out=out)
        am, bm = L.AM_KC, L.AM_KC
    elif kind == "DX":
        ldw = ldw or N
        A = torch.randn(M, K, device=dev)
        Wfull = torch.randn(K, ldw, device=dev)
        sd = dict(A=A.data_ptr(), B=Wfull.data_ptr() + 4 * (ldw - N), C=None, M=M, N=N, K=K, lda=K, ldb=ldw, ldc=N)
        ref = lambda: A.double() @ Wfull[:, ldw - N:].double()
        ven = lambda out: torch.matmul(A, Wfull[:, ldw - N:], out=out)
        am, bm = L.AM_KC, L.AM_RC
    else:
        A = torch.randn(K, M, device=dev)
        X = torch.randn(K, N, device=dev)
        sd = dict(A=A.data_ptr(), B=X.data_ptr(), C=None, M=M, N=N, K=K, lda=M, ldb=N, ldc=N)
        ref = lambda: A.double().t() @ X.double()
        ven = lambda out: torch.matmul(A.t(), X, out=out)
        am, bm = L.AM_RC, L.AM_RC
    Cout = torch.zeros(M, N, device=dev)
    sd["C"] = Cout.data_ptr()
    if general:  # an all-positive ReLU-mask operand routes the launch to the general kernel without changing the result
        aux = torch.ones(A.shape, device=dev)
        aux_keep.append(aux)
        sd["Aaux"] = aux.data_ptr()
    d = P.gemm_descs(ctx, am, bm, L.CM_PLAIN, [sd], 0)[0]
    return d, Cout, ref, ven, (A, ctx)


def time_fn(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    quick = "--quick" in sys.argv
    shapes = [("F", 4096, 1024, 1024, None), ("F", 4096, 1024, 4096, None), ("F", 4096, 1024, 1024, 6157), ("F", 8192, 1024, 1024, None),
              ("F", 4096, 6157, 6157, None), ("DX", 4096, 1024, 1024, 6157), ("DX", 4096, 1024, 1024, None), ("DX", 4096, 6157, 6157, None),
              ("DW", 1024, 1024, 4096, None), ("DW", 1024, 1024, 8192, None), ("DW", 6157, 6157, 4096, None),
              # ragged: M, N not multiples of 128, K with a partial tile
              ("F", 4000, 1000, 1037, None), ("DX", 4090, 1037, 1000, 1100), ("DW", 1000, 1037, 4090, None)]
    custom = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--shape=")]  # --shape=DX,4096,1024,128[,ldw]
    if custom:
        shapes = []
        for c in custom:
            f = c.split(",")
            shapes.append((f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]) if len(f) > 4 else None))
    elif quick:
        shapes = shapes[:2] + shapes[5:6] + shapes[8:9] + shapes[-3:]
    for kind, M, N, K, ldw in shapes:
        d, Cout, ref, ven, keep = desc(kind, M, N, K, ldw)
        L.check(lib.nasrec_launch(st, C.addressof(d)))
        torch.cuda.synchronize()
        r = ref()
        err = float((Cout.double() - r).abs().max() / r.abs().max())
        ms = bench.time_desc(lib, L, st, d, iters=20)
        dg, Cg, _, _, keepg = desc(kind, M, N, K, ldw, general=True)
        msg = bench.time_desc(lib, L, st, dg, iters=10)
        out = torch.empty(M, N, device=dev)
        msv = time_fn(lambda: ven(out))
        fl = 2.0 * M * N * K
        print("%-2s M=%5d N=%5d K=%5d ldw=%-5s splitk=%d  fast %8.1f us = %6.1f TF | general %8.1f us = %6.1f TF | vendor %8.1f us = %6.1f TF | rel.err %.1e" % (
            kind, M, N, K, ldw, d.splitk, ms * 1e3, fl / ms / 1e9, msg * 1e3, fl / msg / 1e9, msv * 1e3, fl / msv / 1e9, err))
        assert err < 5e-6, "wrong result"
        del d, dg, keep, keepg
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
