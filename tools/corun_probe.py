#!/usr/bin/env python3
"""Do a Transformer kernel (vector-ALU bound, one workgroup per sample) and a throughput GEMM (matrix-pipe bound) of a cfg-3 supernet
step overlap when they are launched on two streams?  Times each alone (back to back on one stream) and the pair on two streams; the
question behind running the dense and the sparse branch of a block side by side at large batch (supernet.py:1113-1134)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import bench
from nasrec_amd import _lib as L, plan as P
from nasrec_amd.search_space import ops_config_lib
from nasrec_amd.supernet.supernet import SuperNet
from nasrec_amd.utils.config import DATASETS

w = bench.WORKLOADS[int(os.environ.get("CONFIG", "3"))]
B = w["B"]
lib = L.load()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ds = DATASETS[w["dataset"]]
tables = [min(n, w["cap"]) if w.get("cap") else n for n in ds["tables"]]
bx = bench.synthetic_batches(1, B, ds["Fd"], tables, dev, 1234, zero_dense=(w["dataset"] == "avazu"))[0]
torch.manual_seed(0)
model = SuperNet(num_blocks=7, ops_config=ops_config_lib[w["space"]], use_layernorm=True, num_embeddings=tables, sparse_input_size=ds["Fs"],
                 path_sampling_strategy="full-path", fixed=False, anypath_choice="binomial-0.5").to(dev)
with torch.no_grad():
    model(bx[0][:64], bx[1][:64])
eng = model._engine
eng.init_weights(seed=0)
model.configure_path_sampling_strategy("full-path")
ch = model._resolve_choice(None)
eng.train_step(bx[0], bx[1], bx[2], 1e-3, choice=ch)
torch.cuda.synchronize()
cp = eng.compile(ch, B, train=True)
descs = list(cp.fwd.descs) + list(cp.bwd.descs)
def pick(pred):
    return next(d for d in descs if pred(d))
mha_b = pick(lambda d: d.kind == L.OP_MHA_BWD)
mha_f = pick(lambda d: d.kind == L.OP_MHA_FWD)
gem_small = pick(lambda d: isinstance(d, L.GemmDesc) and P.gemm_kernel_name(d) == "gemm_fast_kernel" and not d.zmode and d.splitk == 1 and d.seg[0].M * d.seg[0].N <= 4096 * 1024 and sum(d.seg[q].K for q in range(d.nseg)) >= 1000)
gem_big = max((d for d in descs if isinstance(d, L.GemmDesc) and P.gemm_kernel_name(d) == "gemm_fast_kernel" and d.splitk == 1), key=bench.gemm_flops)
ln = pick(lambda d: d.kind == L.OP_LAYERNORM_BWD)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def launch(stream, d, n):
    for _ in range(n):
        rc = lib.nasrec_launch(C.c_void_p(stream.cuda_stream), C.byref(d))
        assert rc == 0, lib.nasrec_last_error()
def timed(pairs, n=10):
    """pairs: [(stream, desc)] launched n times each (interleaved enqueue); wall time from a common start to both streams' ends"""
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in pairs]
    for st, d in pairs:
        launch(st, d, 2)
    torch.cuda.synchronize()
    e0.record(torch.cuda.current_stream())
    for st, _ in pairs:
        st.wait_event(e0)
    for _ in range(n):
        for st, d in pairs:
            launch(st, d, 1)
    for (st, _), e in zip(pairs, ends):
        e.record(st)
    torch.cuda.synchronize()
    return max(e0.elapsed_time(e) for e in ends) / n * 1e3
def name(d):
    if isinstance(d, L.GemmDesc):
        return "gemm_fast %dx%dx%d (%d problems)" % (d.seg[0].M, d.seg[0].N, sum(d.seg[q].K for q in range(d.nseg)) if not d.zmode else d.seg[0].K, d.nseg if d.zmode else 1)
    return {L.OP_MHA_BWD: "mha_bwd", L.OP_MHA_FWD: "mha_fwd", L.OP_LAYERNORM_BWD: "layernorm_bwd"}[d.kind]
for a, b in ((mha_b, gem_small), (mha_b, gem_big), (mha_f, gem_small), (ln, gem_small), (mha_b, mha_f)):
    ta, tb = timed([(s1, a)]), timed([(s1, b)])
    tab = timed([(s1, a), (s2, b)])
    tba = timed([(s2, b), (s1, a)])
    print("%-16s %7.1f us | %-44s %7.1f us | side by side %7.1f / %7.1f us (sum %7.1f, max %7.1f)" % (name(a), ta, name(b), tb, tab, tba, ta + tb, max(ta, tb)))
