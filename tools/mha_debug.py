#!/usr/bin/env python3
"""Which intermediate of the fused Transformer first leaves the fp64 reference (saved planes of the forward, then every gradient)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from nasrec_amd import _lib as L
lib = L.load()
B, N, dims = int(os.environ.get("B", 3)), int(os.environ.get("N", 64)), int(os.environ.get("DIMS", -1))
torch.manual_seed(1)
shapes = [(48, 16), (48,), (16, 16), (16,), (16,), (16,), (16, 16), (16,), (16, 16), (16,), (16,), (16,)]
names = ["Win", "bin", "Wout", "bout", "l1w", "l1b", "W1", "c1", "W2", "c2", "l2w", "l2b"]
p = [torch.randn(s) * (0.3 if len(s) == 2 else 0.1) for s in shapes]
p[4] = 0.17 + 0.02 * torch.randn(16); p[10] = 0.17 + 0.02 * torch.randn(16)
x = torch.randn(B, N, 16)
if dims >= 0:
    x[:, dims:] = 0
pd = [t.double().requires_grad_(True) for t in p]
xd = x.double().requires_grad_(True)
Win, bin_, Wout, bout, l1w, l1b, W1, c1, W2, c2, l2w, l2b = pd
qkv = xd @ Win.t() + bin_
q, k, v = qkv[..., :16], qkv[..., 16:32], qkv[..., 32:]
qh, kh, vh = [t.reshape(B, N, 8, 2).permute(0, 2, 1, 3) for t in (q, k, v)]
s = qh @ kh.transpose(-1, -2) / math.sqrt(2)
pm = torch.softmax(s, -1)
o = (pm @ vh).permute(0, 2, 1, 3).reshape(B, N, 16)
r1 = o @ Wout.t() + bout + xd
mu1, var1 = r1.mean(-1, keepdim=True), r1.var(-1, unbiased=False, keepdim=True)
xh1 = (r1 - mu1) / torch.sqrt(var1 + 1e-5); h1 = xh1 * l1w + l1b
f1 = torch.relu(h1 @ W1.t() + c1); r2 = f1 @ W2.t() + c2 + h1
mu2, var2 = r2.mean(-1, keepdim=True), r2.var(-1, unbiased=False, keepdim=True)
xh2 = (r2 - mu2) / torch.sqrt(var2 + 1e-5); out = xh2 * l2w + l2b
if dims >= 0:
    out = out * (torch.arange(N) < dims).double()[None, :, None]
dev = lambda t: t.to("cuda").contiguous()
gp, gx, gout = [dev(t) for t in p], dev(x), dev(torch.zeros(B, N, 16))
saved = dev(torch.zeros(B * N * L.MHA_SAVED))
d = L.MhaDesc()
d.kind, d.B, d.N, d.ldx, d.ldo, d.dims_in_use = L.OP_MHA_FWD, B, N, N * 16, N * 16, dims
d.x, d.out, d.saved = gx.data_ptr(), gout.data_ptr(), saved.data_ptr()
for i in range(12):
    d.params[i] = gp[i].data_ptr()
L.check(lib.nasrec_launch(None, C.addressof(d))); torch.cuda.synchronize()
sv = saved.cpu().double().reshape(B, N * L.MHA_SAVED)
plane = lambda i: sv[:, i * N * 16:(i + 1) * N * 16].reshape(B, N, 16)
def rep(name, a, b):
    print("  %-8s max err %.3e (scale %.2e)" % (name, float((a - b).abs().max()), float(b.abs().max())))
print("forward B=%d N=%d dims=%d" % (B, N, dims))
rep("o", plane(0), o.detach())
m = plane(1)
smax = s.detach().max(-1).values.permute(0, 2, 1); ssum = torch.exp(s.detach() - s.detach().max(-1, keepdim=True).values).sum(-1).permute(0, 2, 1)
rep("max", m[..., :8], smax); rep("1/sum", m[..., 8:], 1 / ssum)
rs = sv[:, 2 * N * 16:2 * N * 16 + N * 4].reshape(B, N, 4)
rep("rstd1", rs[..., 0], (1 / torch.sqrt(var1 + 1e-5)).detach()[..., 0]); rep("rstd2", rs[..., 1], (1 / torch.sqrt(var2 + 1e-5)).detach()[..., 0])
rep("mu1", rs[..., 2], mu1.detach()[..., 0]); rep("mu2", rs[..., 3], mu2.detach()[..., 0])
rep("out", gout.cpu().double(), out.detach())
dout = torch.randn(B, N, 16)
out.backward(dout.double())
gdo, dx, part = dev(dout), dev(torch.zeros(B, N, 16)), dev(torch.zeros(B, L.MHA_PARAMS))
e = L.MhaDesc()
e.kind, e.B, e.N, e.ldx, e.ldo, e.dims_in_use = L.OP_MHA_BWD, B, N, N * 16, N * 16, dims
e.x, e.dout, e.dx, e.dparams_partial, e.saved, e.bwd_form = gx.data_ptr(), gdo.data_ptr(), dx.data_ptr(), part.data_ptr(), saved.data_ptr(), 4
for i in range(12):
    e.params[i] = gp[i].data_ptr()
L.check(lib.nasrec_launch(None, C.addressof(e))); torch.cuda.synchronize()
print("backward")
rep("dx", dx.cpu().double(), xd.grad)
tot = part.cpu().double().sum(0)
off = 0
for nme, sh, t in zip(names, shapes, pd):
    n = int(np.prod(sh))
    rep("d" + nme, tot[off:off + n].reshape(sh), t.grad)
    off += n
