#!/usr/bin/env python3
"""Per-launch table of one sampled-path training step of a supernet bench workload (bench.py --config 3/4/5): every descriptor of
the step timed on its own with HIP events, GEMMs with their problem list and achieved TFLOP/s.  CONFIG=3 SEED=0 TOP=40."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
model.configure_path_sampling_strategy(os.environ.get("STRATEGY", "default"))
np.random.seed(int(os.environ.get("SEED", "0")))
ch = model._resolve_choice(None)
eng.train_step(bx[0], bx[1], bx[2], 1e-3, choice=ch)
torch.cuda.synchronize()
cp = eng.compile(ch, B, train=True)
sp = torch.cuda.current_stream(dev).cuda_stream
names = {getattr(L, n): n[3:] for n in dir(L) if n.startswith("OP_")}
rows, tot = [], 0.0
for phase, prog in (("fwd", cp.fwd), ("bwd", cp.bwd), ("opt", cp.opt)):
    for d in prog.descs:
        us = bench.time_desc(lib, L, sp, d, iters=10) * 1e3
        tot += us
        info = ""
        if isinstance(d, L.GemmDesc):
            segs = [(d.seg[q].M, d.seg[q].N, d.seg[q].K, d.seg[q].ones_col) for q in range(d.nseg) if d.seg[q].A]
            fl = bench.gemm_flops(d)
            info = "%s am=%d bm=%d cm=%d z=%d splitk=%d act=%d %.1f TF/s  %s" % (
                P.gemm_kernel_name(d), d.amode, d.bmode, d.cmode, d.zmode, d.splitk, d.act, fl / us / 1e6,
                " ".join("%dx%dx%d%s" % (s[0], s[1], s[2], "+1" if s[3] else "") for s in segs))
            # epilogue flags + leading dimensions of the first problem (what separates a plain store from the general epilogue)
            s0 = d.seg[0]
            info += "  [%s ld=%d/%d/%d]" % ("".join(c for c, on in (("b", bool(d.bias)), ("p", bool(d.pre_add)), ("z", bool(d.save_z)), ("a", bool(d.save_act)),
                                                                   ("m", d.mul_nseg > 0), ("d", d.dims_in_use >= 0), ("B", d.beta != 0), ("A", bool(s0.accumulate)),
                                                                   ("r", bool(d.rowsum_out) or bool(s0.rowsum))) if on) or "-", s0.lda, s0.ldb, s0.ldc)
        elif isinstance(d, L.MhaDesc):
            info = "N=%d dims=%d" % (d.N, d.dims_in_use)
        elif isinstance(d, L.ReduceRowsDesc):
            info = "R=%d C=%d ld=%d ndst=%d" % (d.R, d.C, d.ld, d.ndst)
        elif isinstance(d, L.LayerNormDesc):
            info = "mode=%d R=%d D=%d" % (d.mode, d.R, d.D)
        rows.append((us, phase, names.get(d.kind, str(d.kind)), info))
print("sum of isolated launches: %.1f us over %d launches" % (tot, len(rows)))
for us, phase, name, info in sorted(rows, reverse=True)[:int(os.environ.get("TOP", "40"))]:
    print("%9.1f us  %s %-14s %s" % (us, phase, name, info))
# per-category totals
cat = {}
for us, phase, name, info in rows:
    key = name + (" " + info.split()[0] if name == "GEMM" else "")
    if name == "GEMM" and " cm=1" in info or "am=3" in info:
        key = "GEMM token-axis"
    c = cat.setdefault(key, [0.0, 0])
    c[0] += us
    c[1] += 1
print("by kind:")
for k, (us, n) in sorted(cat.items(), key=lambda kv: -kv[1][0]):
    print("  %-28s %9.1f us  %4d launches" % (k, us, n))
